"""ctypes binding of librtx_host.so (include/rtx_host.h) and scene construction on top of it.

Host-side only: per-frame Scene::update equivalents, this repo's own BLAS builder, the procedural
cfg3 "atrium" stand-in for the absent Sponza mesh.  Produces pyrtx.scene_io.Scene objects, which
both the product (api.Renderer) and the oracle consume as plain data.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import scene_io as sio
from .ctypes_structs import RtxCamera, RtxTextureDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "host", "librtx_host.so")
EXPORTS = ["rtxh_camera_basis", "rtxh_camera_update", "rtxh_quaternion_axis_angle", "rtxh_scene_dynamic_animate", "rtxh_scene_update", "rtxh_instance_update", "rtxh_plane_update", "rtxh_tlas_create", "rtxh_tlas_destroy",
           "rtxh_tlas_build", "rtxh_blas_build", "rtxh_blas_build_reference_bvh", "rtxh_blas_build_reference_sbvh", "rtxh_texture_mips", "rtxh_texture_load", "rtxh_texture_free", "rtxh_sky_load", "rtxh_image_load", "rtxh_image_free", "rtxh_image_save_png", "rtxh_atrium_generate", "rtxh_mesh_free",
           "rtxh_obj_load", "rtxh_obj_free", "rtxh_mtl_load", "rtxh_bvh_cache_load", "rtxh_bvh_cache_save", "rtxh_bvh_cache_free"]

PI = np.float32(3.14159265359)          # Util.h:8


class RtxhMesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("material_ids", C.POINTER(C.c_int32)), ("triangle_count", C.c_int32), ("material_count", C.c_int32)]


class RtxhObj(C.Structure):
    _fields_ = [("mesh", RtxhMesh), ("materials", C.c_void_p), ("texture_names", C.c_void_p)]


class RtxhBvhCache(C.Structure):
    _fields_ = [("triangle_count", C.c_int32), ("node_count", C.c_int32), ("index_count", C.c_int32), ("pad", C.c_int32),
                ("hot", C.c_void_p), ("cold", C.c_void_p), ("nodes", C.c_void_p), ("indices", C.c_void_p)]


TEXNAME_MAX = 512
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: run __graft_entry__.build()")
        l = C.CDLL(path)
        vp, i32 = C.c_void_p, C.c_int32
        l.rtxh_camera_basis.argtypes = [i32, i32, C.c_float, vp, vp, C.POINTER(RtxCamera)]
        l.rtxh_instance_update.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        l.rtxh_plane_update.argtypes = [vp, vp, i32, vp]
        l.rtxh_tlas_create.argtypes = [i32, C.POINTER(vp)]
        l.rtxh_tlas_destroy.argtypes = [vp]
        l.rtxh_tlas_build.argtypes = [vp, vp, vp, vp, vp, C.POINTER(i32)]
        l.rtxh_blas_build.argtypes = [vp, i32, i32, vp, C.POINTER(i32), vp]
        l.rtxh_blas_build_reference_bvh.argtypes = [vp, i32, vp, C.POINTER(i32), vp]
        l.rtxh_quaternion_axis_angle.argtypes = [vp, C.c_float, vp]
        l.rtxh_camera_update.argtypes = [C.c_float, C.c_uint32, vp, vp]
        l.rtxh_scene_dynamic_animate.argtypes = [C.c_float, C.POINTER(C.c_float), vp, vp, i32]
        l.rtxh_scene_update.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
        l.rtxh_texture_load.argtypes = [C.c_char_p, i32, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int64), C.POINTER(RtxTextureDesc)]
        l.rtxh_texture_free.argtypes = [C.POINTER(C.c_float)]
        l.rtxh_sky_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(i32)]
        l.rtxh_image_load.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.POINTER(C.c_uint8))]
        l.rtxh_image_free.argtypes = [C.POINTER(C.c_uint8)]
        l.rtxh_image_save_png.argtypes = [C.c_char_p, vp, i32, i32]
        l.rtxh_blas_build_reference_sbvh.argtypes = [vp, i32, vp, i32, C.POINTER(i32), vp, i32, C.POINTER(i32)]
        l.rtxh_obj_load.argtypes = [C.c_char_p, C.POINTER(RtxhObj)]
        l.rtxh_obj_free.argtypes = [C.POINTER(RtxhObj)]
        l.rtxh_mtl_load.argtypes = [C.c_char_p, C.POINTER(RtxhObj)]
        l.rtxh_bvh_cache_load.argtypes = [C.c_char_p, C.POINTER(RtxhBvhCache)]
        l.rtxh_bvh_cache_save.argtypes = [C.c_char_p, C.POINTER(RtxhBvhCache)]
        l.rtxh_bvh_cache_free.argtypes = [C.POINTER(RtxhBvhCache)]
        l.rtxh_texture_mips.argtypes = [vp, i32, i32, C.POINTER(RtxTextureDesc), C.POINTER(C.c_int64)]
        l.rtxh_atrium_generate.argtypes = [C.c_uint32, i32, C.POINTER(RtxhMesh)]
        l.rtxh_mesh_free.argtypes = [C.POINTER(RtxhMesh)]
        for n in EXPORTS:
            getattr(l, n).restype = C.c_int
        _lib = l
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def axis_angle(axis: Sequence[float], angle: float) -> np.ndarray:
    """Quaternion::axis_angle, Quaternion.h:24-34 (sinf/cosf in fp32)."""
    out = np.zeros(4, np.float32)
    a = _f32(axis)
    assert lib().rtxh_quaternion_axis_angle(a.ctypes.data, C.c_float(float(np.float32(angle))), out.ctypes.data) == 0
    return out


KEYS = {"W": 1, "A": 2, "S": 4, "D": 8, "LSHIFT": 16, "SPACE": 32, "UP": 64, "DOWN": 128, "LEFT": 256, "RIGHT": 512}   # RTXH_KEY_*


def camera_update(delta: float, keys, position, rotation):
    """Camera::update's keyboard half (Camera.cpp:18-39); keys: iterable of names from KEYS.  Returns (position, rotation)."""
    p, r = _f32(position).copy(), _f32(rotation).copy()
    mask = 0
    for k in keys:
        mask |= KEYS[k]
    assert lib().rtxh_camera_update(C.c_float(float(np.float32(delta))), mask, p.ctypes.data, r.ctypes.data) == 0
    return p, r


def camera_basis(width: int, height: int, fov: float, position, rotation) -> np.ndarray:
    out = RtxCamera()
    p, r = _f32(position), _f32(rotation)
    rc = lib().rtxh_camera_basis(width, height, C.c_float(fov), p.ctypes.data, r.ctypes.data, C.byref(out))
    assert rc == 0
    cam = np.zeros(1, sio.CAMERA)
    C.memmove(cam.ctypes.data, C.byref(out), C.sizeof(out))
    return cam


def instance_update(position, rotation, root_min, root_max, blas_id: int):
    inst = np.zeros(1, sio.INSTANCE)
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    p, r, a, b = _f32(position), _f32(rotation), _f32(root_min), _f32(root_max)
    rc = lib().rtxh_instance_update(p.ctypes.data, r.ctypes.data, a.ctypes.data, b.ctypes.data, inst.ctypes.data, mn.ctypes.data, mx.ctypes.data)
    assert rc == 0
    inst["blas_id"] = blas_id
    return inst, mn, mx


def plane_update(position, rotation, material_id: int) -> np.ndarray:
    out = np.zeros(1, sio.PLANE)
    p, r = _f32(position), _f32(rotation)
    assert lib().rtxh_plane_update(p.ctypes.data, r.ctypes.data, material_id, out.ctypes.data) == 0
    return out


class Tlas:
    """Persistent TLAS builder: TopLevelBVH::init + build_bvh every frame (TopLevelBVH.cpp:5-45)."""

    def __init__(self, n: int):
        self.n = n
        self.h = C.c_void_p()
        assert lib().rtxh_tlas_create(n, C.byref(self.h)) == 0

    def build(self, positions: np.ndarray, aabbs: np.ndarray):
        p, a = _f32(positions).reshape(self.n, 3), _f32(aabbs).reshape(self.n, 6)
        nodes = np.zeros(2 * self.n, sio.BVH_NODE)
        idx = np.zeros(self.n, np.int32)
        nc = C.c_int32()
        assert lib().rtxh_tlas_build(self.h, p.ctypes.data, a.ctypes.data, nodes.ctypes.data, idx.ctypes.data, C.byref(nc)) == 0
        return nodes[:nc.value].copy(), idx

    def __del__(self):
        try:
            lib().rtxh_tlas_destroy(self.h)
        except Exception:
            pass


class DynamicScene:
    """Per-frame state of a scene whose instances move: Scene::update (Scene.cpp:139-171) in two native calls per frame."""

    def __init__(self, scene: sio.Scene, positions, rotations):
        self.n = len(scene.instances)
        self.pos = np.ascontiguousarray(positions, np.float32).reshape(self.n, 3).copy()
        self.rot = np.ascontiguousarray(rotations, np.float32).reshape(self.n, 4).copy()
        self.time = C.c_float(0.0)
        self.blas_ids = np.ascontiguousarray(scene.instances["blas_id"], np.int32)
        self.roots = np.concatenate([np.concatenate([b.nodes[0]["aabb_min"], b.nodes[0]["aabb_max"]]) for b in scene.blas]).astype(np.float32)
        self.tlas = Tlas(self.n)
        self.instances = np.zeros(self.n, sio.INSTANCE)
        self.nodes = np.zeros(2 * self.n, sio.BVH_NODE)
        self.indices = np.zeros(self.n, np.int32)

    def animate_dynamic(self, delta: float):
        assert lib().rtxh_scene_dynamic_animate(C.c_float(delta), C.byref(self.time), self.pos.ctypes.data, self.rot.ctypes.data, self.n) == 0

    def update(self):
        """-> (instances, tlas_nodes, tlas_indices) for rtx_set_frame"""
        nc = C.c_int32()
        rc = lib().rtxh_scene_update(self.tlas.h, self.n, self.pos.ctypes.data, self.rot.ctypes.data, self.blas_ids.ctypes.data, self.roots.ctypes.data,
                                     self.instances.ctypes.data, self.nodes.ctypes.data, self.indices.ctypes.data, C.byref(nc))
        assert rc == 0, rc
        return self.instances, self.nodes[:nc.value], self.indices


def load_obj(path: str):
    """OBJLoader::load_obj equivalent: (positions (n,3,3), normals, texcoords (n,3,2), material ids, MATERIAL array,
    texture path per material or None)."""
    o = RtxhObj()
    rc = lib().rtxh_obj_load(path.encode(), C.byref(o))
    if rc:
        raise RuntimeError(f"rtxh_obj_load({path}) failed: {rc}")
    n, nm = o.mesh.triangle_count, o.mesh.material_count
    pos = np.ctypeslib.as_array(o.mesh.positions, (n, 3, 3)).copy()
    nrm = np.ctypeslib.as_array(o.mesh.normals, (n, 3, 3)).copy()
    uv = np.ctypeslib.as_array(o.mesh.texcoords, (n, 3, 2)).copy()
    mid = np.ctypeslib.as_array(o.mesh.material_ids, (n,)).copy()
    mats = np.frombuffer((C.c_char * (nm * sio.MATERIAL.itemsize)).from_address(o.materials), sio.MATERIAL).copy()
    names_raw = (C.c_char * (nm * TEXNAME_MAX)).from_address(o.texture_names).raw
    names = [names_raw[i * TEXNAME_MAX:(i + 1) * TEXNAME_MAX].split(b"\0")[0].decode() or None for i in range(nm)]
    lib().rtxh_obj_free(C.byref(o))
    return pos, nrm, uv, mid, mats, names


def build_blas(positions: np.ndarray, normals: np.ndarray, texcoords: np.ndarray, material_ids: np.ndarray,
               material_offset: int, bins: int = 32, reference_bvh: bool = False, reference_sbvh: bool = False) -> sio.Blas:
    """Triangle soup (n,3,3) + per-vertex normals (n,3,3) + texcoords (n,3,2) -> flattened BLAS
    (TriangleHot / TriangleCold as OBJLoader.cpp:156-175 fills them, in leaf order as BottomLevelBVH::flatten :196-212)."""
    pos = _f32(positions).reshape(-1, 9)
    n = pos.shape[0]
    nc = C.c_int32()
    if reference_sbvh:    # the reference's default MESH_ACCELERATOR_SBVH topology (BVHBuilders.h:48-329): leaves may share triangles
        cap = 2 * n
        while True:
            nodes = np.zeros(2 * cap, sio.BVH_NODE)
            order = np.zeros(cap, np.int32)
            oc = C.c_int32()
            rc = lib().rtxh_blas_build_reference_sbvh(pos.ctypes.data, n, nodes.ctypes.data, len(nodes), C.byref(nc),
                                                      order.ctypes.data, cap, C.byref(oc))
            if rc != 4 or cap >= 32 * n:   # RTX_ERR_LIMIT: more duplicated references than the reference's own 2n arrays hold
                break                          # (beyond 32n the input is pathological, e.g. piles of coplanar overlapping triangles: give up)
            cap *= 2
        if rc:
            raise ValueError(f"rtxh_blas_build_reference_sbvh failed with status {rc}" + (" (reference explosion: try the plain BVH)" if rc == 4 else ""))
        order = order[:oc.value]
    else:
        nodes = np.zeros(2 * n, sio.BVH_NODE)
        order = np.zeros(n, np.int32)
        if reference_bvh:     # the reference's MESH_ACCELERATOR_BVH topology (BVHBuilders.h:8-46)
            rc = lib().rtxh_blas_build_reference_bvh(pos.ctypes.data, n, nodes.ctypes.data, C.byref(nc), order.ctypes.data)
        else:                 # this repo's binned-SAH builder
            rc = lib().rtxh_blas_build(pos.ctypes.data, n, bins, nodes.ctypes.data, C.byref(nc), order.ctypes.data)
        assert rc == 0, rc
    m = len(order)
    p = pos.reshape(n, 3, 3)[order]
    nr = _f32(normals).reshape(n, 3, 3)[order]
    uv = _f32(texcoords).reshape(n, 3, 2)[order]
    hot = np.zeros(m, sio.TRI_HOT)
    hot["position_0"] = p[:, 0]; hot["position_edge_1"] = p[:, 1] - p[:, 0]; hot["position_edge_2"] = p[:, 2] - p[:, 0]
    cold = np.zeros(m, sio.TRI_COLD)
    cold["tex_coord_0"] = uv[:, 0]; cold["tex_coord_edge_1"] = uv[:, 1] - uv[:, 0]; cold["tex_coord_edge_2"] = uv[:, 2] - uv[:, 0]
    cold["normal_0"] = nr[:, 0]; cold["normal_edge_1"] = nr[:, 1] - nr[:, 0]; cold["normal_edge_2"] = nr[:, 2] - nr[:, 0]
    cold["material_id"] = np.asarray(material_ids, np.int32)[order]
    return sio.Blas(nodes[:nc.value].copy(), hot, cold, material_offset, n)


def load_bvh_cache(path: str, material_offset: int = 0) -> sio.Blas:
    """Reads a `<mesh>.obj.bvh` file exactly as BottomLevelBVH::load_from_disk does (BottomLevelBVH.cpp:171-192:
    int triangle_count; TriangleHot[n]; TriangleCold[n]; int node_count; BVHNode[node_count]; int index_count; int[index_count])
    and applies BottomLevelBVH::flatten (:196-212).  The record layouts are the C ABI's, so no conversion happens."""
    buf = open(path, "rb").read()
    n = int(np.frombuffer(buf, np.int32, 1, 0)[0]); off = 4
    hot = np.frombuffer(buf, sio.TRI_HOT, n, off); off += n * sio.TRI_HOT.itemsize
    cold = np.frombuffer(buf, sio.TRI_COLD, n, off); off += n * sio.TRI_COLD.itemsize
    nc = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    nodes = np.frombuffer(buf, sio.BVH_NODE, nc, off).copy(); off += nc * sio.BVH_NODE.itemsize
    ic = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    idx = np.frombuffer(buf, np.int32, ic, off)
    assert off + 4 * ic == len(buf), "trailing bytes in .bvh cache"
    if nc > 1:
        nodes[1] = np.zeros(1, sio.BVH_NODE)[0]           # never written by the reference (uninitialised heap in the file)
    return sio.Blas(nodes, hot[idx].copy(), cold[idx].copy(), material_offset, n)


def load_bvh_cache_native(path: str, material_offset: int = 0, keep_node1: bool = False) -> sio.Blas:
    """The same file through librtx_host's rtxh_bvh_cache_load (the C / C++ route), flattened like BottomLevelBVH::flatten."""
    c = RtxhBvhCache()
    rc = lib().rtxh_bvh_cache_load(path.encode(), C.byref(c))
    if rc:
        raise ValueError(f"rtxh_bvh_cache_load({path!r}) failed with status {rc}")
    def arr(ptr, dtype, n):
        return np.frombuffer((C.c_char * (n * dtype.itemsize)).from_address(ptr), dtype, n).copy()
    hot = arr(c.hot, sio.TRI_HOT, c.triangle_count); cold = arr(c.cold, sio.TRI_COLD, c.triangle_count)
    nodes = arr(c.nodes, sio.BVH_NODE, c.node_count); idx = arr(c.indices, np.dtype(np.int32), c.index_count)
    n = c.triangle_count
    lib().rtxh_bvh_cache_free(C.byref(c))
    if len(nodes) > 1 and not keep_node1:
        nodes[1] = np.zeros(1, sio.BVH_NODE)[0]
    return sio.Blas(nodes, hot[idx].copy(), cold[idx].copy(), material_offset, n)


def save_bvh_cache(path: str, blas: sio.Blas) -> None:
    """Writes a flattened BLAS in the reference's `.bvh` layout (BottomLevelBVH.cpp:149-168) with an identity index list,
    so the reference's load_from_disk + flatten reproduce the same arrays."""
    n = len(blas.tri_hot)
    with open(path, "wb") as f:
        f.write(np.int32(n).tobytes()); f.write(np.ascontiguousarray(blas.tri_hot).tobytes()); f.write(np.ascontiguousarray(blas.tri_cold).tobytes())
        f.write(np.int32(len(blas.nodes)).tobytes()); f.write(np.ascontiguousarray(blas.nodes).tobytes())
        f.write(np.int32(n).tobytes()); f.write(np.arange(n, dtype=np.int32).tobytes())


def texture_with_mips(level0_rgb: np.ndarray) -> sio.Texture:
    """level0_rgb: (h, w, 3) linear-space float32 -> Texture with the reference's box-filter chain."""
    h, w, _ = level0_rgb.shape
    buf = np.zeros((w * h + (w * h) // 3 + 2, 3), np.float32)
    buf[:w * h] = level0_rgb.reshape(-1, 3)
    desc = RtxTextureDesc()
    cnt = C.c_int64()
    assert lib().rtxh_texture_mips(buf.ctypes.data, w, h, C.byref(desc), C.byref(cnt)) == 0
    d = np.zeros(1, sio.TEXTURE_DESC)
    C.memmove(d.ctypes.data, C.byref(desc), C.sizeof(desc))
    return sio.Texture(d, buf[:cnt.value].copy())


def load_image(path: str) -> np.ndarray:
    """PNG / TGA file -> (h, w, 4) uint8 RGBA, decoded like stbi_load(..., STBI_rgb_alpha) at Texture.cpp:40."""
    w = C.c_int32(); h = C.c_int32(); p = C.POINTER(C.c_uint8)()
    rc = lib().rtxh_image_load(path.encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc:
        raise ValueError(f"rtxh_image_load({path!r}) failed with status {rc}")
    out = np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    lib().rtxh_image_free(p)
    return out


def load_sky(path: str) -> np.ndarray:
    """Sky::Sky (Sky.cpp:8-26): raw float3 probe file -> (size, size, 3) float32."""
    p = C.POINTER(C.c_float)(); n = C.c_int32()
    rc = lib().rtxh_sky_load(path.encode(), C.byref(p), C.byref(n))
    if rc:
        raise ValueError(f"rtxh_sky_load({path!r}) failed with status {rc}")
    out = np.ctypeslib.as_array(p, (n.value, n.value, 3)).copy()
    lib().rtxh_texture_free(p)
    return out


def save_png(path: str, packed: np.ndarray) -> None:
    """(h, w) uint32 0x00RRGGBB frame -> RGB PNG."""
    p = np.ascontiguousarray(packed, np.uint32)
    rc = lib().rtxh_image_save_png(path.encode(), p.ctypes.data, p.shape[1], p.shape[0])
    if rc:
        raise ValueError(f"rtxh_image_save_png({path!r}) failed with status {rc}")


def load_texture(path: str, mipmap_mode: bool = True) -> sio.Texture:
    """Texture::load (Texture.cpp:30-129): decode, sRGB -> linear, box-filter mips when both sides are powers of two."""
    p = C.POINTER(C.c_float)(); cnt = C.c_int64(); desc = RtxTextureDesc()
    rc = lib().rtxh_texture_load(path.encode(), 1 if mipmap_mode else 0, C.byref(p), C.byref(cnt), C.byref(desc))
    if rc:
        raise ValueError(f"rtxh_texture_load({path!r}) failed with status {rc}")
    texels = np.ctypeslib.as_array(p, (cnt.value, 3)).copy()
    lib().rtxh_texture_free(p)
    d = np.zeros(1, sio.TEXTURE_DESC)
    C.memmove(d.ctypes.data, C.byref(desc), C.sizeof(desc))
    return sio.Texture(d, texels)


_SRGB_LUT = None


def srgb8_to_linear(img_u8: np.ndarray) -> np.ndarray:
    """Texture::load's decode (Texture.cpp:13-20,64-72): byte * 0.00392156862f, then Math::gamma_to_linear
    (Math.h:78-88) with the host libm's powf — evaluated once per byte value into a 256-entry table."""
    global _SRGB_LUT
    if _SRGB_LUT is None:
        libm = C.CDLL("libm.so.6")
        libm.powf.restype = C.c_float; libm.powf.argtypes = [C.c_float, C.c_float]
        lut = np.zeros(256, np.float32)
        for b in range(256):
            x = np.float32(b) * np.float32(0.00392156862)
            if x <= 0: v = np.float32(0.0)
            elif x >= 1: v = np.float32(1.0)
            elif x < np.float32(0.04045): v = x / np.float32(12.92)
            else: v = np.float32(libm.powf(float((x + np.float32(0.055)) / np.float32(1.055)), 2.4))
            lut[b] = v
        _SRGB_LUT = lut
    return _SRGB_LUT[img_u8]


def synthetic_sky(n: int = 64) -> np.ndarray:
    """Seeded stand-in for Data/Sky_Probes/rnl_probe.float (absent from the reference mount);
    identical to the probe the golden vectors were rendered with (oracle/ref_harness/make_goldens.py)."""
    y, x = np.mgrid[0:n, 0:n]
    sky = np.zeros((n, n, 3), np.float32)
    sky[..., 0] = 0.5 + 0.5 * x / n
    sky[..., 1] = 0.6 + 0.2 * (((x // 8) + (y // 8)) & 1)
    sky[..., 2] = 0.5 + 0.5 * y / n
    return sky


def procedural_texture_images(seed: int) -> List[np.ndarray]:
    """The cfg3 stand-in's four power-of-two textures as 8-bit sRGB images (h, w, 3) — the form an asset file has."""
    rng = np.random.RandomState(seed & 0x7fffffff)

    def noise(n, octaves=4):
        out = np.zeros((n, n), np.float32)
        for o in range(octaves):
            k = 2 ** (o + 2)
            g = rng.rand(k, k).astype(np.float32)
            out += np.kron(g, np.ones((n // k, n // k), np.float32)) / (2 ** o)
        return out / out.max()

    texs = []
    n = 256   # floor tiles
    y, x = np.mgrid[0:n, 0:n]
    tile = (((x // 32) + (y // 32)) & 1).astype(np.float32)
    grout = ((x % 32 < 2) | (y % 32 < 2)).astype(np.float32)
    base = 0.55 + 0.25 * tile - 0.3 * grout + 0.1 * noise(n)
    texs.append(np.stack([base, base * 0.95, base * 0.85], -1))
    n = 512   # bricks
    y, x = np.mgrid[0:n, 0:n]
    row = y // 32
    xs = (x + (row & 1) * 32) % 64
    mortar = ((xs < 3) | (y % 32 < 3)).astype(np.float32)
    nb = noise(n)
    texs.append(np.stack([0.62 - 0.35 * mortar + 0.15 * nb, 0.34 - 0.1 * mortar + 0.1 * nb, 0.27 - 0.05 * mortar + 0.08 * nb], -1))
    n = 256   # stone
    s = 0.45 + 0.4 * noise(n, 5)
    texs.append(np.stack([s, s * 0.97, s * 0.9], -1))
    n = 256   # fabric stripes
    y, x = np.mgrid[0:n, 0:n]
    f = 0.6 + 0.3 * np.sin(x * (2 * np.pi / 16)).astype(np.float32) + 0.08 * noise(n)
    texs.append(np.stack([f, f, f], -1))
    return [np.clip(np.rint(np.clip(t, 0, 1) * 255.0), 0, 255).astype(np.uint8) for t in texs]


def _procedural_textures(seed: int) -> List[sio.Texture]:
    return [texture_with_mips(srgb8_to_linear(img)) for img in procedural_texture_images(seed)]


def make_config(width, height, bounces, mip_filter=1, texture_mode=2, stack_size=64, traversal=1, max_aniso=8.0, heatmap=0) -> np.ndarray:
    cfg = np.zeros(1, sio.CONFIG)
    cfg["width"] = width; cfg["height"] = height; cfg["bounces"] = bounces; cfg["stack_size"] = stack_size
    cfg["traversal_strategy"] = traversal; cfg["texture_mode"] = texture_mode; cfg["mip_filter"] = mip_filter
    cfg["max_anisotropy"] = max_aniso; cfg["heatmap"] = heatmap
    return cfg


ATRIUM_CAMERA = ((24.0, 5.5, 1.2), ((0.0, 1.0, 0.0), -1.45))     # position, (axis, angle)
ATRIUM_POINT = ((30.0, 26.0, 20.0), (-6.0, 9.0, 0.5))            # colour, position
ATRIUM_SPOT = ((40.0, 40.0, 48.0), (20.0, 14.0, -2.0), (-0.8, -0.55, 0.25), 50.0, 70.0)   # colour, position, direction, inner, outer (degrees)
ATRIUM_DIR = ((0.9, 0.9, 0.9), (0.1, -1.0, 0.1))                 # Scene.cpp:125


def atrium_mesh(seed: int = 0x5EED0003, detail: int = 1):
    """(positions (n,3,3), normals (n,3,3), texcoords (n,3,2) [v already flipped], material ids (n,), material count)."""
    mesh = RtxhMesh()
    assert lib().rtxh_atrium_generate(seed, detail, C.byref(mesh)) == 0
    n = mesh.triangle_count
    pos = np.ctypeslib.as_array(mesh.positions, (n, 3, 3)).copy()
    nrm = np.ctypeslib.as_array(mesh.normals, (n, 3, 3)).copy()
    uv = np.ctypeslib.as_array(mesh.texcoords, (n, 3, 2)).copy()
    mid = np.ctypeslib.as_array(mesh.material_ids, (n,)).copy()
    nmat = mesh.material_count
    lib().rtxh_mesh_free(C.byref(mesh))
    return pos, nrm, uv, mid, nmat


def atrium_materials(nmat: int) -> np.ndarray:
    """25 mesh-local materials mimicking sponza.mtl (Kd ~0.47, a few with Ks / Kt), texture ids into procedural_texture_images."""
    m = np.zeros(nmat, sio.MATERIAL)
    m["texture_id"] = -1; m["index_of_refraction"] = 1.0
    m["diffuse"] = 0.4704
    kd = {0: (0.75, 0.75, 0.75), 1: (0.8, 0.8, 0.8), 2: (0.7, 0.68, 0.6), 3: (0.8, 0.78, 0.74), 4: (0.6, 0.58, 0.5), 5: (0.75, 0.73, 0.7), 6: (0.5, 0.47, 0.4)}
    for k, v in kd.items():
        m["diffuse"][k] = v
    for k, v in {0: 0, 1: 1, 2: 1, 3: 2, 4: 2, 5: 2, 6: 2}.items():
        m["texture_id"][k] = v
    m["reflection"][0] = (0.15, 0.15, 0.15)                       # polished floor
    curtain = [(0.7, 0.1, 0.1), (0.1, 0.5, 0.15), (0.1, 0.15, 0.7), (0.7, 0.6, 0.1), (0.6, 0.1, 0.6), (0.1, 0.6, 0.6)]
    for i, c in enumerate(curtain):
        m["diffuse"][7 + i] = c; m["texture_id"][7 + i] = 3
    # vases: glass, metal, tinted glass, dark metal
    m["diffuse"][13] = (0.05, 0.05, 0.05); m["reflection"][13] = (0.25, 0.25, 0.25); m["transmittance"][13] = (0.95, 0.95, 0.95); m["index_of_refraction"][13] = 1.5
    m["diffuse"][14] = (0.3, 0.25, 0.1); m["reflection"][14] = (0.8, 0.7, 0.4)
    m["diffuse"][15] = (0.05, 0.1, 0.05); m["reflection"][15] = (0.2, 0.2, 0.2); m["transmittance"][15] = (0.6, 0.9, 0.7); m["index_of_refraction"][15] = 1.33
    m["diffuse"][16] = (0.1, 0.1, 0.12); m["reflection"][16] = (0.6, 0.6, 0.65)
    for i in range(4):                                               # statues: bronze-ish
        m["diffuse"][17 + i] = (0.45 - 0.05 * i, 0.3, 0.15 + 0.04 * i); m["reflection"][17 + i] = (0.12, 0.1, 0.06)
    m["diffuse"][21] = (0.35, 0.3, 0.25); m["diffuse"][22] = (0.6, 0.6, 0.6); m["diffuse"][23] = (0.85, 0.85, 0.8); m["diffuse"][24] = (0.5, 0.5, 0.5)
    return m


def atrium_scene(width: int = 1920, height: int = 1080, bounces: int = 3, detail: int = 1, seed: int = 0x5EED0003,
                 mip_filter: int = 1, bins: int = 32, accel: str = "sbvh") -> sio.Scene:
    """BASELINE.json configs[2]: Sponza-class (255 296 triangles at detail 1) stand-in, 3 lights,
    reflect/refract depth 3, anisotropic mip filter (the shipped default, Config.h:53).
    accel: "sbvh" = the reference's default builder restated (MESH_ACCELERATOR_SBVH, Config.h:35), "bvh" = its
    non-spatial builder, "binned" = this repo's own binned-SAH builder."""
    assert accel in ("sbvh", "bvh", "binned"), accel
    pos, nrm, uv, mid, nmat = atrium_mesh(seed, detail)
    sc = sio.Scene()
    sc.config = make_config(width, height, bounces, mip_filter=mip_filter)
    sc.textures = _procedural_textures(seed)
    # material table: 0 = MaterialBuffer default (Material.h:52-60), then the mesh's materials
    mats = np.zeros(1 + nmat, sio.MATERIAL)
    mats["texture_id"] = -1; mats["index_of_refraction"] = 1.0
    mats[1:] = atrium_materials(nmat)
    sc.materials = mats
    sc.blas = [build_blas(pos, nrm, uv, mid, material_offset=1, bins=bins, reference_sbvh=accel == "sbvh", reference_bvh=accel == "bvh")]
    root = sc.blas[0].nodes[0]
    inst, mn, mx = instance_update((0, 0, 0), (0, 0, 0, 1), root["aabb_min"], root["aabb_max"], 0)
    sc.instances = inst
    tl = Tlas(1)
    sc.tlas_nodes, sc.tlas_indices = tl.build(np.zeros((1, 3), np.float32), np.concatenate([mn, mx])[None])
    sc.sky = synthetic_sky()

    def normalize(v):                                                # Vector3::normalize, Vector3.h:24-31
        v = np.asarray(v, np.float32)
        inv = np.float32(1.0) / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dtype=np.float32)
        return v * inv
    deg2rad = lambda a: np.float32(a) * PI * np.float32(0.00555555555)   # DEG_TO_RAD, Util.h:14
    sc.dir_lights = np.zeros(1, sio.DIR_LIGHT); sc.dir_lights["colour"] = ATRIUM_DIR[0]; sc.dir_lights["negative_direction"] = -normalize(ATRIUM_DIR[1])
    sc.point_lights = np.zeros(1, sio.POINT_LIGHT); sc.point_lights["colour"] = ATRIUM_POINT[0]; sc.point_lights["position"] = ATRIUM_POINT[1]
    sc.spot_lights = np.zeros(1, sio.SPOT_LIGHT); sc.spot_lights["colour"] = ATRIUM_SPOT[0]; sc.spot_lights["position"] = ATRIUM_SPOT[1]
    sc.spot_lights["negative_direction"] = -normalize(ATRIUM_SPOT[2])
    sc.spot_lights["inner_cutoff"] = np.cos(deg2rad(np.float32(0.5) * np.float32(ATRIUM_SPOT[3])), dtype=np.float32)      # SpotLight.h:13-14
    sc.spot_lights["outer_cutoff"] = np.cos(deg2rad(np.float32(0.5) * np.float32(ATRIUM_SPOT[4])), dtype=np.float32)
    fov = float(deg2rad(110.0))                                                                                              # Scene.cpp:75
    sc.camera = camera_basis(width, height, fov, ATRIUM_CAMERA[0], axis_angle(*ATRIUM_CAMERA[1]))
    return sc
