"""ctypes binding of librtx_hip.so (the C ABI declared in include/rtx.h).

This is plumbing for tests and bench.py; the product is the shared library.  There is no CPU
fallback: if the HIP library is missing or no GPU is present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from . import scene_io as sio
from .ctypes_structs import RtxConfig, RtxFrame, RtxStats, RtxWork, RtxTextureDesc, fill_frame

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "librtx_hip.so")

EXPORTS = ["rtx_abi_version", "rtx_create", "rtx_destroy", "rtx_last_error", "rtx_upload_blas", "rtx_upload_materials",
           "rtx_upload_texture", "rtx_upload_sky", "rtx_set_frame", "rtx_render_tiles", "rtx_synchronize", "rtx_get_stats",
           "rtx_read_framebuffer", "rtx_framebuffer_device_ptrs", "rtx_last_kernel_times", "rtx_enable_kernel_timing",
           "rtx_bind_framebuffer", "rtx_set_stream", "rtx_present", "rtx_debug_libm", "rtx_debug_texture_sample", "rtx_debug_sky_sample", "rtx_debug_blas_wide", "rtx_debug_blas_wide_closest",
           "rtx_group_unique_id", "rtx_group_create", "rtx_group_render", "rtx_group_destroy", "rtx_group_attach", "rtx_group_create_local", "rtx_group_render_local",
           "rtx_group_layout", "rtx_group_slot_pixels", "rtx_debug_group_loopback", "rtx_debug_trace_rays", "rtx_debug_occluded", "rtx_debug_light_plot"]

RTX_RENDER_COUNT_WORK = 1
RTX_RENDER_SIMPLE_TRACE = 2
RTX_RENDER_CULL_DEAD_SHADOW_RAYS = 4
RTX_RENDER_SERIAL = 8
RTX_RENDER_LANE_TRACE = 16
RTX_RENDER_PACKET_STATS = 32
RTX_RENDER_PACKET_CLOSEST = 64
ERRORS = {1: "RTX_ERR_INVALID_ARG", 2: "RTX_ERR_NO_DEVICE", 3: "RTX_ERR_HIP", 4: "RTX_ERR_LIMIT", 5: "RTX_ERR_STATE", 6: "RTX_ERR_OOM"}

_lib = None


class RtxError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: {ERRORS.get(code, code)} {detail}")


def load_library(path: Optional[str] = None):
    """Load librtx_hip.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(path or os.environ.get("RTX_HIP_LIB") or LIB_PATH)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: the HIP extension has not been built (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path)
    vp, i32, u32, f = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
    lib.rtx_abi_version.restype = C.c_int
    lib.rtx_create.argtypes = [C.POINTER(RtxConfig), C.POINTER(vp)]
    lib.rtx_destroy.argtypes = [vp]
    lib.rtx_last_error.argtypes = [vp]; lib.rtx_last_error.restype = C.c_char_p
    lib.rtx_upload_blas.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32]
    lib.rtx_upload_materials.argtypes = [vp, vp, i32]
    lib.rtx_upload_texture.argtypes = [vp, i32, C.POINTER(RtxTextureDesc), vp, C.c_int64]
    lib.rtx_upload_sky.argtypes = [vp, vp, i32]
    lib.rtx_set_frame.argtypes = [vp, C.POINTER(RtxFrame)]
    lib.rtx_render_tiles.argtypes = [vp, i32, i32, i32, u32]
    lib.rtx_synchronize.argtypes = [vp]
    lib.rtx_get_stats.argtypes = [vp, C.POINTER(RtxStats), C.POINTER(RtxWork)]
    lib.rtx_read_framebuffer.argtypes = [vp, vp, vp]
    lib.rtx_framebuffer_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.rtx_last_kernel_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(f), i32, C.POINTER(i32)]
    lib.rtx_enable_kernel_timing.argtypes = [vp, i32]
    lib.rtx_bind_framebuffer.argtypes = [vp, vp, vp]
    lib.rtx_set_stream.argtypes = [vp, vp]
    lib.rtx_present.argtypes = [vp, i32, vp, C.POINTER(vp)]
    lib.rtx_debug_libm.argtypes = [vp, i32, vp, vp, vp, i32]
    lib.rtx_debug_texture_sample.argtypes = [vp, i32, vp, vp, i32]
    lib.rtx_debug_sky_sample.argtypes = [vp, vp, vp, i32]
    lib.rtx_debug_blas_wide.argtypes = [vp, i32, vp]
    lib.rtx_debug_blas_wide_closest.argtypes = [vp, i32, vp]
    lib.rtx_group_unique_id.argtypes = [vp]
    lib.rtx_group_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    lib.rtx_group_attach.argtypes = [vp, vp, C.POINTER(vp)]
    lib.rtx_group_render.argtypes = [vp, u32]
    lib.rtx_group_destroy.argtypes = [vp]
    lib.rtx_group_create_local.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
    lib.rtx_group_render_local.argtypes = [C.POINTER(vp), i32, u32]
    lib.rtx_debug_group_loopback.argtypes = [vp, i32, u32]
    lib.rtx_group_layout.argtypes = [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.rtx_group_slot_pixels.argtypes = [i32, i32, i32, i32, vp, C.c_int64]
    lib.rtx_debug_trace_rays.argtypes = [vp, vp, i32, vp, u32]
    lib.rtx_debug_occluded.argtypes = [vp, vp, i32, vp, u32]
    lib.rtx_debug_light_plot.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("rtx_last_error",):
            fn.restype = C.c_int
    _lib = lib
    return lib


def group_layout(width: int, height: int, world: int, rank: int):
    """(tiles_total, tiles_per_rank, own_tiles) of rank `rank` of `world` (rtx_group_layout: host function, no GPU needed).  The rank
    renders rtx_render_tiles(first_tile=rank, tile_stride=world, tile_count=own_tiles)."""
    lib = load_library()
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.rtx_group_layout(width, height, world, rank, C.byref(a), C.byref(b), C.byref(c))
    if rc:
        raise RtxError(rc, "rtx_group_layout")
    return a.value, b.value, c.value


def group_slot_pixels(width: int, height: int, world: int, rank: int) -> np.ndarray:
    """Raster pixel index of every slot of the rank's tile-major send buffer, -1 = padding (rtx_group_slot_pixels: the kernels' own index maths, on the host)."""
    lib = load_library()
    _, per_rank, _ = group_layout(width, height, world, rank)
    out = np.empty(per_rank * 1024, np.int64)
    rc = lib.rtx_group_slot_pixels(width, height, world, rank, out.ctypes.data, out.size)
    if rc:
        raise RtxError(rc, "rtx_group_slot_pixels")
    return out


class Renderer:
    """One rtx_ctx: uploads a Scene and renders tiles of it on one GPU."""

    def __init__(self, scene: sio.Scene, device: int = 0, upload: bool = True):
        self.lib = load_library()
        self.scene = scene
        cfg = RtxConfig()
        C.memmove(C.byref(cfg), scene.config.ctypes.data, C.sizeof(RtxConfig))
        cfg.device = device
        self.ctx = C.c_void_p()
        rc = self.lib.rtx_create(C.byref(cfg), C.byref(self.ctx))
        if rc:
            self.ctx = None
            raise RtxError(rc, "rtx_create")
        self._keep: List[np.ndarray] = []
        if upload:
            self.upload_scene(scene)
            self.set_frame(scene)

    def _chk(self, rc: int, where: str):
        if rc:
            raise RtxError(rc, where, (self.lib.rtx_last_error(self.ctx) or b"").decode())

    def close(self):
        if self.ctx and getattr(self, "group", None):
            self.group_destroy()
        if self.ctx:
            self.lib.rtx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, sc: sio.Scene):
        for i, b in enumerate(sc.blas):
            nodes = np.ascontiguousarray(b.nodes); hot = np.ascontiguousarray(b.tri_hot); cold = np.ascontiguousarray(b.tri_cold)
            self._chk(self.lib.rtx_upload_blas(self.ctx, i, nodes.ctypes.data, len(nodes), hot.ctypes.data, cold.ctypes.data,
                                               len(hot), b.material_offset), "rtx_upload_blas")
        mats = np.ascontiguousarray(sc.materials)
        self._chk(self.lib.rtx_upload_materials(self.ctx, mats.ctypes.data, len(mats)), "rtx_upload_materials")
        for i, t in enumerate(sc.textures):
            desc = RtxTextureDesc()
            C.memmove(C.byref(desc), t.desc.ctypes.data, C.sizeof(RtxTextureDesc))
            tex = np.ascontiguousarray(t.texels, np.float32)
            self._chk(self.lib.rtx_upload_texture(self.ctx, i, C.byref(desc), tex.ctypes.data, len(tex)), "rtx_upload_texture")
        sky = np.ascontiguousarray(sc.sky, np.float32)
        self._chk(self.lib.rtx_upload_sky(self.ctx, sky.ctypes.data, sky.shape[0]), "rtx_upload_sky")

    def set_frame(self, sc: sio.Scene):
        fr = RtxFrame()
        keep: List[np.ndarray] = []
        fill_frame(fr, sc, keep)
        self._chk(self.lib.rtx_set_frame(self.ctx, C.byref(fr)), "rtx_set_frame")

    def render_async(self, first_tile: int = 0, tile_stride: int = 1, tile_count: Optional[int] = None, count_work: bool = False,
                     simple_trace: bool = False, cull_dead_shadow_rays: bool = False, serial: bool = False, lane_trace: bool = False,
                     packet_stats: bool = False, packet_closest: bool = False):
        if tile_count is None:
            tile_count = (self.scene.tile_count - first_tile + tile_stride - 1) // tile_stride
        flags = ((RTX_RENDER_COUNT_WORK if count_work else 0) | (RTX_RENDER_SIMPLE_TRACE if simple_trace else 0)
                 | (RTX_RENDER_CULL_DEAD_SHADOW_RAYS if cull_dead_shadow_rays else 0) | (RTX_RENDER_SERIAL if serial else 0)
                 | (RTX_RENDER_LANE_TRACE if lane_trace else 0) | (RTX_RENDER_PACKET_STATS if packet_stats else 0)
                 | (RTX_RENDER_PACKET_CLOSEST if packet_closest else 0))
        self._chk(self.lib.rtx_render_tiles(self.ctx, first_tile, tile_stride, tile_count, flags), "rtx_render_tiles")

    def synchronize(self):
        self._chk(self.lib.rtx_synchronize(self.ctx), "rtx_synchronize")

    def stats(self):
        st, wk = RtxStats(), RtxWork()
        self._chk(self.lib.rtx_get_stats(self.ctx, C.byref(st), C.byref(wk)), "rtx_get_stats")
        return st.as_dict(), wk.as_dict()

    def framebuffer(self):
        sc = self.scene
        rgb = np.zeros((sc.height, sc.width, 3), np.float32)
        packed = np.zeros((sc.height, sc.width), np.uint32)
        self._chk(self.lib.rtx_read_framebuffer(self.ctx, rgb.ctypes.data, packed.ctypes.data), "rtx_read_framebuffer")
        return rgb, packed

    def present(self, fxaa: bool = True) -> np.ndarray:
        """Window::draw_quad: the displayed image (gamma 1/2.2, optional FXAA) of the frame in the framebuffer, (h, w) 0x00RRGGBB."""
        out = np.zeros((self.scene.height, self.scene.width), np.uint32)
        self._chk(self.lib.rtx_present(self.ctx, 1 if fxaa else 0, out.ctypes.data, None), "rtx_present")
        return out

    def device_ptrs(self):
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.rtx_framebuffer_device_ptrs(self.ctx, C.byref(a), C.byref(b)), "rtx_framebuffer_device_ptrs")
        return a.value, b.value

    def render(self, first_tile: int = 0, tile_stride: int = 1, tile_count: Optional[int] = None, count_work: bool = False,
               simple_trace: bool = False, cull_dead_shadow_rays: bool = False, serial: bool = False, lane_trace: bool = False,
               packet_stats: bool = False, packet_closest: bool = False) -> Dict:
        self.render_async(first_tile, tile_stride, tile_count, count_work, simple_trace, cull_dead_shadow_rays, serial, lane_trace, packet_stats, packet_closest)
        stats, work = self.stats()
        rgb, packed = self.framebuffer()
        return {"rgb": rgb, "packed": packed, "stats": stats, "work": work}

    def bind_framebuffer(self, rgb_ptr: Optional[int], packed_ptr: Optional[int]):
        self._chk(self.lib.rtx_bind_framebuffer(self.ctx, rgb_ptr, packed_ptr), "rtx_bind_framebuffer")

    def set_stream(self, stream_handle: Optional[int]):
        self._chk(self.lib.rtx_set_stream(self.ctx, stream_handle), "rtx_set_stream")

    def enable_timing(self, on: bool = True):
        self._chk(self.lib.rtx_enable_kernel_timing(self.ctx, 1 if on else 0), "rtx_enable_kernel_timing")

    def kernel_times(self):
        cap = 65536
        names = (C.c_char_p * cap)(); ms = (C.c_float * cap)(); n = C.c_int32()
        self._chk(self.lib.rtx_last_kernel_times(self.ctx, names, ms, cap, C.byref(n)), "rtx_last_kernel_times")
        return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, cap))]

    # ---- GPU groups (include/rtx.h: rtx_group_*) ----------------------------------------------------------------
    @staticmethod
    def group_unique_id() -> bytes:
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.rtx_group_unique_id(buf)
        if rc:
            raise RtxError(rc, "rtx_group_unique_id")
        return buf.raw

    def group_create(self, rank: int, world: int, unique_id: Optional[bytes] = None):
        self.group = C.c_void_p()
        self._chk(self.lib.rtx_group_create(self.ctx, rank, world, unique_id, C.byref(self.group)), "rtx_group_create")

    def group_attach(self, base: "Renderer"):
        """Another frame in flight on the same GPU: joins `base`'s communicator as the same rank (rtx_group_attach)."""
        self.group = C.c_void_p()
        self._chk(self.lib.rtx_group_attach(self.ctx, base.group, C.byref(self.group)), "rtx_group_attach")

    def group_render(self, cull_dead_shadow_rays: bool = False, lane_trace: bool = False):
        flags = (RTX_RENDER_CULL_DEAD_SHADOW_RAYS if cull_dead_shadow_rays else 0) | (RTX_RENDER_LANE_TRACE if lane_trace else 0)
        self._chk(self.lib.rtx_group_render(self.group, flags), "rtx_group_render")

    def group_destroy(self):
        if getattr(self, "group", None):
            self.lib.rtx_group_destroy(self.group); self.group = None

    def group_loopback(self, world: int, flags: int = 0):
        self._chk(self.lib.rtx_debug_group_loopback(self.ctx, world, flags), "rtx_debug_group_loopback")

    # ---- unit-level hooks ---------------------------------------------------------------------------
    def debug_libm(self, fn: int, a: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else a
        out = np.zeros_like(a)
        self._chk(self.lib.rtx_debug_libm(self.ctx, fn, a.ctypes.data, bb.ctypes.data, out.ctypes.data, a.size), "rtx_debug_libm")
        return out

    def debug_texture_sample(self, texture_id: int, in6: np.ndarray) -> np.ndarray:
        in6 = np.ascontiguousarray(in6, np.float32)
        out = np.zeros((in6.shape[0], 3), np.float32)
        self._chk(self.lib.rtx_debug_texture_sample(self.ctx, texture_id, in6.ctypes.data, out.ctypes.data, in6.shape[0]), "rtx_debug_texture_sample")
        return out

    def debug_trace_rays(self, rays18: np.ndarray, flags: int = 0) -> np.ndarray:
        rays18 = np.ascontiguousarray(rays18, np.float32)
        out = np.zeros((rays18.shape[0], 27), np.float32)
        self._chk(self.lib.rtx_debug_trace_rays(self.ctx, rays18.ctypes.data, rays18.shape[0], out.ctypes.data, flags), "rtx_debug_trace_rays")
        return out

    def debug_blas_wide(self, blas_id: int) -> int:
        need = C.c_int32(0)
        self._chk(self.lib.rtx_debug_blas_wide(self.ctx, blas_id, C.byref(need)), "rtx_debug_blas_wide")
        return int(need.value)

    def debug_blas_wide_closest(self, blas_id: int) -> int:
        need = C.c_int32(0)
        self._chk(self.lib.rtx_debug_blas_wide_closest(self.ctx, blas_id, C.byref(need)), "rtx_debug_blas_wide_closest")
        return int(need.value)

    def debug_occluded(self, origin_direction_maxdist7: np.ndarray, flags: int = 0) -> np.ndarray:
        r = np.ascontiguousarray(origin_direction_maxdist7, np.float32)
        out = np.zeros(r.shape[0], np.uint32)
        self._chk(self.lib.rtx_debug_occluded(self.ctx, r.ctypes.data, r.shape[0], out.ctypes.data, flags), "rtx_debug_occluded")
        return out

    def debug_light_plot(self, pl, sl, dl, in10: np.ndarray, rgb: np.ndarray):
        in10 = np.ascontiguousarray(in10, np.float32); rgb = np.ascontiguousarray(rgb, np.float32)
        out9 = np.zeros((in10.shape[0], 9), np.float32); packed = np.zeros(rgb.shape[0], np.uint32)
        self._chk(self.lib.rtx_debug_light_plot(self.ctx, pl.ctypes.data, sl.ctypes.data, dl.ctypes.data, in10.ctypes.data, out9.ctypes.data, in10.shape[0],
                                                rgb.ctypes.data, packed.ctypes.data, rgb.shape[0]), "rtx_debug_light_plot")
        return out9, packed

    def debug_sky_sample(self, dirs: np.ndarray) -> np.ndarray:
        dirs = np.ascontiguousarray(dirs, np.float32)
        out = np.zeros_like(dirs)
        self._chk(self.lib.rtx_debug_sky_sample(self.ctx, dirs.ctypes.data, out.ctypes.data, dirs.shape[0]), "rtx_debug_sky_sample")
        return out
