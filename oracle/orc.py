"""ctypes binding of oracle/liboracle.so.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", "cpu-raytracer_amd"))
from pyrtx import scene_io as sio  # noqa: E402
from pyrtx.ctypes_structs import (RtxConfig, RtxFrame, RtxStats, RtxWork, RtxTextureDesc, fill_frame)  # noqa: E402


class OrcBlas(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("tri_hot", C.c_void_p), ("tri_cold", C.c_void_p),
                ("node_count", C.c_int32), ("triangle_count", C.c_int32), ("material_offset", C.c_int32), ("pad", C.c_int32)]


class OrcTexture(C.Structure):
    _fields_ = [("desc", RtxTextureDesc), ("texels", C.c_void_p)]


class OrcScene(C.Structure):
    _fields_ = [("config", RtxConfig), ("frame", RtxFrame),
                ("blas", C.POINTER(OrcBlas)), ("blas_count", C.c_int32),
                ("materials", C.c_void_p), ("material_count", C.c_int32),
                ("textures", C.POINTER(OrcTexture)), ("texture_count", C.c_int32),
                ("sky", C.c_void_p), ("sky_size", C.c_int32)]


_lib = None


def build() -> str:
    path = os.path.join(_HERE, "liboracle.so")
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return path


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_render_tiles.restype = C.c_int
        _lib.orc_render_tiles.argtypes = [C.POINTER(OrcScene), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(RtxStats), C.POINTER(RtxWork), C.c_int32]
        _lib.orc_texture_sample.restype = None
        _lib.orc_texture_sample.argtypes = [C.POINTER(OrcTexture), C.c_int32, C.c_int32, C.c_float] + [C.c_float] * 6 + [C.c_void_p]
        _lib.orc_plot.restype = C.c_uint32
        _lib.orc_plot.argtypes = [C.c_void_p]
        _lib.orc_float_to_int.restype = C.c_int32
        _lib.orc_float_to_int.argtypes = [C.c_float]
        _lib.orc_sky_sample.restype = None
        _lib.orc_sky_sample.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        _lib.orc_aabb_intersect.restype = C.c_int
        _lib.orc_aabb_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        _lib.orc_trace_closest.restype = None
        _lib.orc_trace_closest.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_void_p]
        _lib.orc_trace_any.restype = C.c_int
        _lib.orc_trace_any.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_void_p, C.c_float]
        _lib.orc_calc_lighting.restype = None
        _lib.orc_calc_lighting.argtypes = [C.c_void_p] * 6 + [C.c_float, C.c_void_p]
        _lib.orc_ewa_table.restype = None
        _lib.orc_ewa_table.argtypes = [C.c_void_p]
        _lib.orc_present.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        _lib.orc_present.restype = C.c_int
    return _lib


class OracleScene:
    """Holds the numpy arrays alive and exposes an orc_scene struct over them."""

    def __init__(self, sc: sio.Scene):
        self.sc = sc
        self._keep = []
        s = OrcScene()
        C.memmove(C.byref(s.config), sc.config.ctypes.data, C.sizeof(RtxConfig))
        fill_frame(s.frame, sc, self._keep)
        blas = (OrcBlas * max(1, len(sc.blas)))()
        for i, b in enumerate(sc.blas):
            for a in (b.nodes, b.tri_hot, b.tri_cold):
                assert a.flags["C_CONTIGUOUS"]
            blas[i].nodes = b.nodes.ctypes.data
            blas[i].tri_hot = b.tri_hot.ctypes.data
            blas[i].tri_cold = b.tri_cold.ctypes.data
            blas[i].node_count = len(b.nodes)
            blas[i].triangle_count = len(b.tri_hot)
            blas[i].material_offset = b.material_offset
        s.blas = blas; s.blas_count = len(sc.blas)
        self._mats = np.ascontiguousarray(sc.materials)
        s.materials = self._mats.ctypes.data; s.material_count = len(self._mats)
        tex = (OrcTexture * max(1, len(sc.textures)))()
        self._tex_arrays = []
        for i, t in enumerate(sc.textures):
            C.memmove(C.byref(tex[i].desc), t.desc.ctypes.data, C.sizeof(RtxTextureDesc))
            arr = np.ascontiguousarray(t.texels, dtype=np.float32)
            self._tex_arrays.append(arr)
            tex[i].texels = arr.ctypes.data
        s.textures = tex; s.texture_count = len(sc.textures)
        size = sc.sky.shape[0]
        # one texel of padding: Sky::sample clamps the index to size*size inclusive (Sky.cpp:45)
        self._sky = np.zeros((size * size + 1, 3), np.float32)
        self._sky[:size * size] = sc.sky.reshape(-1, 3)
        s.sky = self._sky.ctypes.data; s.sky_size = size
        self._blas, self._tex, self.struct = blas, tex, s

    def render(self, first_tile: int = 0, tile_stride: int = 1, tile_count: Optional[int] = None, threads: int = 1,
               want_dist: bool = False):
        sc = self.sc
        if tile_count is None:
            tile_count = sc.tile_count
        rgb = np.zeros((sc.height, sc.width, 3), np.float32)
        packed = np.zeros((sc.height, sc.width), np.uint32)
        dist = np.zeros((sc.height, sc.width), np.float32) if want_dist else None
        stats, work = RtxStats(), RtxWork()
        rc = lib().orc_render_tiles(C.byref(self.struct), first_tile, tile_stride, tile_count, rgb.ctypes.data, packed.ctypes.data,
                                    dist.ctypes.data if want_dist else None, C.byref(stats), C.byref(work), threads)
        if rc != 0:
            raise RuntimeError(f"orc_render_tiles failed: {rc}")
        out = {"rgb": rgb, "packed": packed, "stats": stats.as_dict(), "work": work.as_dict()}
        if want_dist:
            out["dist"] = dist
        return out


def gamma_lut() -> np.ndarray:
    """pow(i / 255, 1 / 2.2) per 8-bit value with the host libm's powf (what rtx_present evaluates on the host, too)."""
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float; libm.powf.argtypes = [C.c_float, C.c_float]
    e = np.float32(1.0) / np.float32(2.2)
    return np.array([libm.powf(float(np.float32(i) / np.float32(255.0)), float(e)) for i in range(256)], np.float32)


def present(packed: np.ndarray, fxaa: bool = True) -> np.ndarray:
    """Window::draw_quad restated (parity unpinned vs GL, see rt_oracle.h): packed (h, w) uint32 -> displayed (h, w) uint32."""
    p = np.ascontiguousarray(packed, np.uint32)
    h, w = p.shape
    out = np.zeros((h, w), np.uint32)
    lut = gamma_lut()
    rc = lib().orc_present(p.ctypes.data, w, h, 1 if fxaa else 0, lut.ctypes.data, out.ctypes.data)
    assert rc == 0, rc
    return out
