"""The C-ABI libraries load and export every symbol their headers declare (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from util import REPO


def declared(header):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtxh?_[a-z0-9_]+)\s*\(", text)))


def test_rtx_h_symbols_exported():
    from pyrtx import api
    lib = api.load_library()
    names = declared("rtx.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(api.EXPORTS) == names
    assert lib.rtx_abi_version() == 1


def test_rtx_host_h_symbols_exported():
    from pyrtx import host
    lib = host.lib()
    names = declared("rtx_host.h")
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(host.EXPORTS) == names


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: rtx_create reports RTX_ERR_NO_DEVICE when there is no HIP device."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyrtx import api, host
    sc_cfg = host.make_config(64, 64, 0)
    ctx = C.c_void_p()
    cfg = api.RtxConfig()
    C.memmove(C.byref(cfg), sc_cfg.ctypes.data, C.sizeof(cfg))
    assert api.load_library().rtx_create(C.byref(cfg), C.byref(ctx)) == 2
    assert not ctx.value


def test_struct_sizes_match_the_reference_layouts():
    from pyrtx import scene_io as sio
    assert sio.BVH_NODE.itemsize == 32      # BVHNode, SURVEY.md §8
    assert sio.TRI_HOT.itemsize == 36       # TriangleHot
    assert sio.TRI_COLD.itemsize == 64      # TriangleCold
