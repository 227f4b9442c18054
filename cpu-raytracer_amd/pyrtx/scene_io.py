"""Scene container and the RTXS scene file format (this repo's own format).

A scene is everything the render path consumes after ``Scene::update`` (reference
Scene.cpp:139-171): config knobs, camera basis, lights, materials, textures, sky, flattened
BLAS arrays (BottomLevelBVH.h:5-34), instances, TLAS, spheres and planes.  Record layouts are
the C structs of include/rtx.h, expressed here as numpy dtypes.

File layout (little endian): 8-byte magic ``RTXSCN01`` then chunks
``tag[4] | n:u32 | nbytes:u64 | payload | pad to 8``.
"""
from __future__ import annotations

import io
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

MAGIC = b"RTXSCN01"
RTX_MAX_MIP_LEVELS = 16
TILE = 32

f32, i32, u32, u64 = np.float32, np.int32, np.uint32, np.uint64

CONFIG = np.dtype([("width", i32), ("height", i32), ("bounces", i32), ("stack_size", i32),
                   ("traversal_strategy", i32), ("texture_mode", i32), ("mip_filter", i32),
                   ("max_anisotropy", f32), ("device", i32), ("heatmap", i32), ("reserved", i32, (6,))])
BVH_NODE = np.dtype([("aabb_min", f32, (3,)), ("aabb_max", f32, (3,)), ("left_or_first", i32), ("count", i32)])
TRI_HOT = np.dtype([("position_0", f32, (3,)), ("position_edge_1", f32, (3,)), ("position_edge_2", f32, (3,))])
TRI_COLD = np.dtype([("tex_coord_0", f32, (2,)), ("tex_coord_edge_1", f32, (2,)), ("tex_coord_edge_2", f32, (2,)),
                     ("normal_0", f32, (3,)), ("normal_edge_1", f32, (3,)), ("normal_edge_2", f32, (3,)),
                     ("material_id", i32)])
MATERIAL = np.dtype([("diffuse", f32, (3,)), ("texture_id", i32), ("reflection", f32, (3,)),
                     ("transmittance", f32, (3,)), ("index_of_refraction", f32), ("pad", i32)])
TEXTURE_DESC = np.dtype([("width", i32), ("height", i32), ("mipmapped", i32), ("mip_levels", i32),
                         ("mip_offsets", i32, (RTX_MAX_MIP_LEVELS,))])
INSTANCE = np.dtype([("blas_id", i32), ("pad", i32, (3,)), ("world", f32, (16,)), ("world_inv", f32, (16,))])
SPHERE = np.dtype([("center", f32, (3,)), ("radius_inv", f32), ("radius_squared", f32), ("material_id", i32)])
PLANE = np.dtype([("normal", f32, (3,)), ("distance", f32), ("u_axis", f32, (3,)), ("v_axis", f32, (3,)),
                  ("material_id", i32), ("pad", i32)])
POINT_LIGHT = np.dtype([("colour", f32, (3,)), ("position", f32, (3,))])
SPOT_LIGHT = np.dtype([("colour", f32, (3,)), ("position", f32, (3,)), ("negative_direction", f32, (3,)),
                       ("inner_cutoff", f32), ("outer_cutoff", f32)])
DIR_LIGHT = np.dtype([("colour", f32, (3,)), ("negative_direction", f32, (3,))])
CAMERA = np.dtype([("position", f32, (3,)), ("rotated_top_left_corner", f32, (3,)),
                   ("rotated_x_axis", f32, (3,)), ("rotated_y_axis", f32, (3,))])

assert CONFIG.itemsize == 64 and BVH_NODE.itemsize == 32 and TRI_HOT.itemsize == 36 and TRI_COLD.itemsize == 64
assert MATERIAL.itemsize == 48 and TEXTURE_DESC.itemsize == 80 and INSTANCE.itemsize == 144
assert SPHERE.itemsize == 24 and PLANE.itemsize == 48 and CAMERA.itemsize == 48


@dataclass
class Blas:
    nodes: np.ndarray        # BVH_NODE[n]
    tri_hot: np.ndarray      # TRI_HOT[m]  (leaf order: flattened)
    tri_cold: np.ndarray     # TRI_COLD[m]
    material_offset: int = 0
    source_triangle_count: int = 0


@dataclass
class Texture:
    desc: np.ndarray         # TEXTURE_DESC scalar array (shape (1,))
    texels: np.ndarray       # float32 (n, 3)


@dataclass
class Scene:
    config: np.ndarray = field(default_factory=lambda: np.zeros(1, CONFIG))
    camera: np.ndarray = field(default_factory=lambda: np.zeros(1, CAMERA))
    ambient: np.ndarray = field(default_factory=lambda: np.full(3, 0.2, f32))      # Scene.h:33
    point_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, POINT_LIGHT))
    spot_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, SPOT_LIGHT))
    dir_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, DIR_LIGHT))
    materials: np.ndarray = field(default_factory=lambda: np.zeros(1, MATERIAL))
    textures: List[Texture] = field(default_factory=list)
    sky: np.ndarray = field(default_factory=lambda: np.zeros((1, 1, 3), f32))       # (size, size, 3)
    blas: List[Blas] = field(default_factory=list)
    instances: np.ndarray = field(default_factory=lambda: np.zeros(0, INSTANCE))
    tlas_nodes: np.ndarray = field(default_factory=lambda: np.zeros(0, BVH_NODE))
    tlas_indices: np.ndarray = field(default_factory=lambda: np.zeros(0, i32))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, SPHERE))
    planes: np.ndarray = field(default_factory=lambda: np.zeros(0, PLANE))
    extra: Dict[str, np.ndarray] = field(default_factory=dict)   # IPOS / IAAB etc. (not consumed by the renderer)

    # -- convenience -----------------------------------------------------------------------------
    @property
    def width(self) -> int: return int(self.config["width"][0])
    @property
    def height(self) -> int: return int(self.config["height"][0])
    @property
    def tile_count_x(self) -> int: return (self.width + TILE - 1) // TILE      # Window.cpp:11
    @property
    def tile_count_y(self) -> int: return (self.height + TILE - 1) // TILE     # Window.cpp:12
    @property
    def tile_count(self) -> int: return self.tile_count_x * self.tile_count_y

    def triangle_refs(self) -> int:
        return sum(len(b.tri_hot) for b in self.blas)


def _chunks(buf: bytes):
    if buf[:8] != MAGIC:
        raise ValueError("not an RTXS scene file")
    pos = 8
    while pos < len(buf):
        tag = buf[pos:pos + 4].decode("ascii")
        n, nbytes = struct.unpack_from("<IQ", buf, pos + 4)
        pos += 16
        payload = buf[pos:pos + nbytes]
        pos += (nbytes + 7) & ~7
        yield tag, n, payload


def load_scene(path_or_bytes) -> Scene:
    if isinstance(path_or_bytes, (bytes, bytearray)):
        buf = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            buf = fh.read()
    sc = Scene()
    blas_hdr: Dict[int, np.ndarray] = {}
    blas_parts: Dict[int, Dict[str, np.ndarray]] = {}
    tex_desc: Dict[int, np.ndarray] = {}
    tex_data: Dict[int, np.ndarray] = {}
    for tag, n, payload in _chunks(buf):
        fb = lambda dt: np.frombuffer(payload, dtype=dt).copy()
        if tag == "CONF": sc.config = fb(CONFIG)
        elif tag == "CAMR": sc.camera = fb(CAMERA)
        elif tag == "AMBT": sc.ambient = fb(f32)[:3]
        elif tag == "LPNT": sc.point_lights = fb(POINT_LIGHT)
        elif tag == "LSPT": sc.spot_lights = fb(SPOT_LIGHT)
        elif tag == "LDIR": sc.dir_lights = fb(DIR_LIGHT)
        elif tag == "MATL": sc.materials = fb(MATERIAL)
        elif tag == "TEXH": tex_desc[n] = fb(TEXTURE_DESC)
        elif tag == "TEXL": tex_data[n] = fb(f32).reshape(-1, 3)
        elif tag == "SKYD": sc.sky = fb(f32).reshape(n, n, 3)
        elif tag == "BLSH": blas_hdr[n] = fb(i32)
        elif tag == "BLSN": blas_parts.setdefault(n, {})["nodes"] = fb(BVH_NODE)
        elif tag == "BLST": blas_parts.setdefault(n, {})["hot"] = fb(TRI_HOT)
        elif tag == "BLSC": blas_parts.setdefault(n, {})["cold"] = fb(TRI_COLD)
        elif tag == "INST": sc.instances = fb(INSTANCE)
        elif tag == "TLSN": sc.tlas_nodes = fb(BVH_NODE)
        elif tag == "TLSI": sc.tlas_indices = fb(i32)
        elif tag == "SPHR": sc.spheres = fb(SPHERE)
        elif tag == "PLAN": sc.planes = fb(PLANE)
        else: sc.extra[tag] = fb(np.uint8)
    for b in sorted(blas_hdr):
        h = blas_hdr[b]
        p = blas_parts[b]
        assert len(p["nodes"]) == h[0] and len(p["hot"]) == h[1] and len(p["cold"]) == h[1]
        sc.blas.append(Blas(p["nodes"], p["hot"], p["cold"], int(h[2]), int(h[3])))
    for t in sorted(tex_desc):
        sc.textures.append(Texture(tex_desc[t], tex_data[t]))
    return sc


def save_scene(sc: Scene, path) -> None:
    out = io.BytesIO()
    out.write(MAGIC)

    def chunk(tag: str, n: int, arr) -> None:
        data = np.ascontiguousarray(arr).tobytes()
        out.write(tag.encode("ascii"))
        out.write(struct.pack("<IQ", n, len(data)))
        out.write(data)
        out.write(b"\0" * ((8 - len(data) % 8) % 8))

    chunk("CONF", 1, sc.config)
    chunk("CAMR", 1, sc.camera)
    chunk("AMBT", 1, np.concatenate([np.asarray(sc.ambient, f32), np.zeros(1, f32)]))
    chunk("LPNT", len(sc.point_lights), sc.point_lights)
    chunk("LSPT", len(sc.spot_lights), sc.spot_lights)
    chunk("LDIR", len(sc.dir_lights), sc.dir_lights)
    chunk("MATL", len(sc.materials), sc.materials)
    for t, tex in enumerate(sc.textures):
        chunk("TEXH", t, tex.desc)
        chunk("TEXL", t, tex.texels.astype(f32))
    chunk("SKYD", sc.sky.shape[0], sc.sky.astype(f32))
    for b, bl in enumerate(sc.blas):
        chunk("BLSH", b, np.array([len(bl.nodes), len(bl.tri_hot), bl.material_offset, bl.source_triangle_count], i32))
        chunk("BLSN", b, bl.nodes)
        chunk("BLST", b, bl.tri_hot)
        chunk("BLSC", b, bl.tri_cold)
    chunk("INST", len(sc.instances), sc.instances)
    chunk("TLSN", len(sc.tlas_nodes), sc.tlas_nodes)
    chunk("TLSI", len(sc.tlas_indices), sc.tlas_indices)
    chunk("SPHR", len(sc.spheres), sc.spheres)
    chunk("PLAN", len(sc.planes), sc.planes)
    for tag, arr in sc.extra.items():
        chunk(tag, 0, arr)
    with open(path, "wb") as fh:
        fh.write(out.getvalue())


def plot_quantise(rgb: np.ndarray) -> np.ndarray:
    """Window::plot (Window.h:56-65) on a (..., 3) float32 array -> packed 0x00RRGGBB uint32.
    np.rint is round-half-to-even, the same as cvtss2si under the default MXCSR mode."""
    c = np.clip(rgb.astype(f32) * f32(255.0), f32(0.0), f32(255.0)) - f32(0.5)
    q = np.rint(c).astype(np.int32)
    return ((q[..., 0] << 16) | (q[..., 1] << 8) | q[..., 2]).astype(np.uint32)
