"""Scene assembly from asset files: what the reference's Scene constructor + Scene::update do on the host before the
first render_tile (Scene.cpp:5-171), built only from this repo's host library — no reference-generated data involved.

  MeshLibrary.load      BottomLevelBVH::load (BottomLevelBVH.cpp:18-58): per-file cache, `.bvh` cache file next to the OBJ,
                        OBJLoader::load_obj / load_mtl (OBJLoader.cpp:8-187), SBVH / BVH build, flatten; materials are
                        appended to the global table behind the default material (MaterialBuffer, Material.h:28-61),
                        textures come from Texture::load's per-path cache (Texture.cpp:11,31-34).
  scene_from_script     the small scene description language of oracle/ref_harness/refdump.cpp (size / bounces / mesh /
                        point / spot / dir / camera / ambient ...), so the same text drives the real reference and this path.
"""
import os
from typing import Dict, List, Optional

import numpy as np

from . import host
from . import scene_io as sio


class MeshLibrary:
    def __init__(self, accel: str = "sbvh", texture_mode: int = 2, use_bvh_cache: bool = False):
        assert accel in ("sbvh", "bvh", "binned")
        self.accel = accel
        self.mipmap_mode = texture_mode == 2                       # TEXTURE_SAMPLE_MODE_MIPMAP, Config.h:38-42
        self.use_bvh_cache = use_bvh_cache
        self.blas: List[sio.Blas] = []
        self.blas_by_file: Dict[str, int] = {}
        self.textures: List[sio.Texture] = []
        self.texture_by_file: Dict[str, int] = {}
        m = np.zeros(1, sio.MATERIAL)                              # MaterialBuffer::init: index 0 = default material, Material.h:52-60
        m["texture_id"] = -1; m["index_of_refraction"] = 1.0
        self.materials = m

    def reserve_material(self) -> int:
        """Primitive::material_id = MaterialBuffer::reserve() (Primitive.h:6) with Material's defaults (Material.h:7-14)."""
        m = np.zeros(1, sio.MATERIAL)
        m["diffuse"] = 1.0; m["texture_id"] = -1; m["index_of_refraction"] = 1.0
        self.materials = np.concatenate([self.materials, m])
        return len(self.materials) - 1

    def texture(self, path: str) -> int:
        if path not in self.texture_by_file:
            self.texture_by_file[path] = len(self.textures)
            self.textures.append(host.load_texture(path, self.mipmap_mode))
        return self.texture_by_file[path]

    def load(self, obj_path: str) -> int:
        """Returns the BLAS id for this file (shared by every instance of it)."""
        if obj_path in self.blas_by_file:
            return self.blas_by_file[obj_path]
        offset = len(self.materials)
        pos, nrm, uv, mid, mats, names = host.load_obj(obj_path)
        mats = mats.copy()
        for i, name in enumerate(names):
            mats["texture_id"][i] = self.texture(name) if name is not None else -1
        self.materials = np.concatenate([self.materials, mats])
        cache = obj_path + ".bvh"
        if self.use_bvh_cache and os.path.exists(cache):
            b = host.load_bvh_cache(cache, offset)
        else:
            b = host.build_blas(pos, nrm, uv, mid, material_offset=offset, reference_sbvh=self.accel == "sbvh", reference_bvh=self.accel == "bvh")
            if self.use_bvh_cache:
                host.save_bvh_cache(cache, b)
        self.blas_by_file[obj_path] = len(self.blas)
        self.blas.append(b)
        return self.blas_by_file[obj_path]


def _normalize(v):                                                  # Vector3::normalize, Vector3.h:24-31
    v = np.asarray(v, np.float32)
    inv = np.float32(1.0) / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dtype=np.float32)
    return v * inv


def _deg2rad(a):
    return np.float32(a) * host.PI * np.float32(0.00555555555)     # DEG_TO_RAD, Util.h:14


def scene_from_script(text: str, data_dir: str, accel: str = "sbvh", mip_filter: int = 1, texture_mode: int = 2,
                      library: Optional[MeshLibrary] = None) -> sio.Scene:
    """`mesh ./Data/X.obj ...` paths are resolved against data_dir's parent the way the reference resolves them against
    its working directory (Util.h:6), i.e. pass the directory that plays the role of ./Data."""
    lib = library or MeshLibrary(accel, texture_mode)
    W, H, bounces = 256, 256, 3
    cam_pos, cam_rot = (0.0, 0.0, 0.0), np.array([0, 0, 0, 1], np.float32)
    meshes, points, spots, dirs, spheres, planes, edits = [], [], [], [], [], [], []
    ambient = None
    ident = np.array([0, 0, 0, 1], np.float32)
    for line in text.splitlines():
        w = line.split("#")[0].split()
        if not w:
            continue
        f = lambda i: np.float32(float(w[i]))
        if w[0] == "size": W, H = int(w[1]), int(w[2])
        elif w[0] == "bounces": bounces = int(w[1])
        elif w[0] == "camera": cam_pos = (f(1), f(2), f(3)); cam_rot = np.array([f(4), f(5), f(6), f(7)], np.float32)
        elif w[0] == "camera_axis_angle": cam_pos = (f(1), f(2), f(3)); cam_rot = host.axis_angle((f(4), f(5), f(6)), f(7))
        elif w[0] == "ambient": ambient = np.array([f(1), f(2), f(3)], np.float32)
        elif w[0] == "mesh": meshes.append((w[1], (f(2), f(3), f(4)), np.array([f(5), f(6), f(7), f(8)], np.float32) if len(w) >= 9 else np.array([0, 0, 0, 1], np.float32)))
        elif w[0] == "mesh_axis_angle": meshes.append((w[1], (f(2), f(3), f(4)), host.axis_angle((f(5), f(6), f(7)), f(8))))
        elif w[0] == "point": points.append(((f(1), f(2), f(3)), (f(4), f(5), f(6))))
        elif w[0] == "spot": spots.append(((f(1), f(2), f(3)), (f(4), f(5), f(6)), (f(7), f(8), f(9)), f(10), f(11)))
        elif w[0] == "dir": dirs.append(((f(1), f(2), f(3)), (f(4), f(5), f(6))))
        elif w[0] == "sphere": spheres.append(((f(1), f(2), f(3)), f(4)))
        elif w[0] == "plane": planes.append(((f(1), f(2), f(3)), np.array([f(4), f(5), f(6), f(7)], np.float32) if len(w) >= 8 else ident))
        elif w[0] == "plane_axis_angle": planes.append(((f(1), f(2), f(3)), host.axis_angle((f(4), f(5), f(6)), f(7))))
        elif w[0] == "matset": edits.append((w[1], w[2], w[3:]))
        elif w[0] in ("nolights", "texprobe"): pass                # texprobe: a harness-only dump of Texture::sample values
        else: raise ValueError(f"scene script command not supported here: {w[0]}")
    if not meshes:
        raise ValueError("script defines no mesh (TopLevelBVH::init asserts count > 0)")

    def data_path(path):
        return os.path.join(data_dir, path[len("./Data/"):] if path.startswith("./Data/") else path)

    sc = sio.Scene()
    sc.config = host.make_config(W, H, bounces, mip_filter=mip_filter, texture_mode=texture_mode)
    sc.spheres = np.zeros(len(spheres), sio.SPHERE)
    for i, (c, r) in enumerate(spheres):                            # Sphere::init, Sphere.h:13-16
        sc.spheres["center"][i] = c; sc.spheres["radius_inv"][i] = np.float32(1.0) / r; sc.spheres["radius_squared"][i] = r * r
        sc.spheres["material_id"][i] = lib.reserve_material()
    sc.planes = np.zeros(len(planes), sio.PLANE)
    for i, (p, q) in enumerate(planes):                             # Plane::update, Plane.cpp:3-11
        sc.planes[i] = host.plane_update(p, q, lib.reserve_material())[0]
    n = len(meshes)
    inst = np.zeros(n, sio.INSTANCE); pos = np.zeros((n, 3), np.float32); aabb = np.zeros((n, 6), np.float32)
    sc_blas_ids = []
    for i, (path, p, q) in enumerate(meshes):
        bid = lib.load(data_path(path))
        root = lib.blas[bid].nodes[0]
        one, mn, mx = host.instance_update(p, q, root["aabb_min"], root["aabb_max"], bid)     # Mesh::update, Mesh.cpp:9-15
        inst[i] = one[0]; pos[i] = p; aabb[i] = np.concatenate([mn, mx]); sc_blas_ids.append(bid)
    for target, field, args in edits:                               # the harness's `matset` (edits MaterialBuffer::materials in place)
        t = target.split(":")
        if t[0] == "mesh": mid = lib.blas[int(sc_blas_ids[int(t[1])])].material_offset + int(t[2])
        elif t[0] == "sphere": mid = int(sc.spheres["material_id"][int(t[1])])
        elif t[0] == "plane": mid = int(sc.planes["material_id"][int(t[1])])
        elif t[0] == "global": mid = int(t[1])
        else: raise ValueError(target)
        m = lib.materials
        if field in ("diffuse", "reflection", "transmittance"): m[field][mid] = [np.float32(float(a)) for a in args[:3]]
        elif field == "ior": m["index_of_refraction"][mid] = np.float32(float(args[0]))
        elif field == "notexture": m["texture_id"][mid] = -1
        elif field == "texture": m["texture_id"][mid] = lib.texture(data_path(args[0]))
        else: raise ValueError(field)
    sc.blas = lib.blas; sc.materials = lib.materials; sc.textures = lib.textures
    sc.instances = inst
    sc.tlas_nodes, sc.tlas_indices = host.Tlas(n).build(pos, aabb)                               # TopLevelBVH::build_bvh, TopLevelBVH.cpp:32-45
    sc.sky = host.synthetic_sky()
    if ambient is not None:
        sc.ambient = ambient
    sc.point_lights = np.zeros(len(points), sio.POINT_LIGHT)
    for i, (c, p) in enumerate(points):
        sc.point_lights["colour"][i] = c; sc.point_lights["position"][i] = p
    sc.spot_lights = np.zeros(len(spots), sio.SPOT_LIGHT)
    for i, (c, p, d, inner, outer) in enumerate(spots):                                           # SpotLight.h:9-15
        sc.spot_lights["colour"][i] = c; sc.spot_lights["position"][i] = p; sc.spot_lights["negative_direction"][i] = -_normalize(d)
        sc.spot_lights["inner_cutoff"][i] = np.cos(_deg2rad(np.float32(0.5) * inner), dtype=np.float32)
        sc.spot_lights["outer_cutoff"][i] = np.cos(_deg2rad(np.float32(0.5) * outer), dtype=np.float32)
    sc.dir_lights = np.zeros(len(dirs), sio.DIR_LIGHT)
    for i, (c, d) in enumerate(dirs):
        sc.dir_lights["colour"][i] = c; sc.dir_lights["negative_direction"][i] = -_normalize(d)
    sc.camera = host.camera_basis(W, H, float(_deg2rad(110.0)), cam_pos, cam_rot)                # fov: Scene.cpp:75
    return sc
