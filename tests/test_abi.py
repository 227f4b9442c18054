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


def test_missing_extension_fails_loudly(tmp_path):
    """No silent fallback: when librtx_hip.so cannot be found the product path raises; nothing under oracle/ is imported by it."""
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['RTX_HIP_LIB'] = %r\n"
            "from pyrtx import api\n"
            "try:\n    api.load_library()\nexcept Exception as e:\n    print('RAISED', type(e).__name__); sys.exit(0)\nsys.exit(1)\n") % (
        os.path.join(REPO, "cpu-raytracer_amd"), str(tmp_path / "nowhere" / "librtx_hip.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "RAISED" in out.stdout, (out.stdout, out.stderr)
    # and the product package never reaches for the checker
    for root, _, files in os.walk(os.path.join(REPO, "cpu-raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert not re.search(r"^\s*(import orc|from orc )|#\s*include[^\n]*rt_oracle|liboracle\.so", text, re.M), os.path.join(root, f)


def test_integration_md_quotes_the_compiled_binding_verbatim():
    """INTEGRATION.md's GpuBackend.cpp is the file `make -C oracle/ref_harness binding_check` compiles against the reference's headers."""
    import os
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = open(os.path.join(repo, "cpu-raytracer_amd", "host", "reference_binding", "GpuBackend.cpp")).read()
    assert code in open(os.path.join(repo, "INTEGRATION.md")).read()
    mk = open(os.path.join(repo, "oracle", "ref_harness", "Makefile")).read()
    assert "binding_check:" in mk and "reference_binding/GpuBackend.cpp" in mk
