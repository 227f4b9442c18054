"""csrc/rtx_libm.h (the device's acosf / atanf / atan2f / expf / log2f) bit-for-bit against the host libm.
CPU part: the header compiled for the host, sampled here (the exhaustive 2^32 sweep is
tests/native/libm_check.c with stride 1: ~10 s per function on 8 cores, all five pass).  GPU part: the device code."""
import os
import subprocess

import numpy as np
import pytest

from util import REPO


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("libm") / "libm_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(REPO, "tests", "native", "libm_check.c"), "-lm", "-lpthread"])
    return exe


@pytest.mark.parametrize("fn,arg", [("acosf", 13), ("atanf", 13), ("atan2f", 30000000), ("expf", 7), ("log2f", 7)])
def test_libm_ports_match_host_libm(checker, fn, arg):
    # acosf / atanf: every 13th of ALL 2^32 bit patterns (330 M values each); stride 1 is the exhaustive sweep quoted in DESIGN.md §3
    out = subprocess.run([checker, fn, str(arg)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


@pytest.mark.gpu
def test_device_libm_matches_host():
    import ctypes as C
    import util
    from pyrtx import api
    sc, _ = util.load_golden("cube")
    r = api.Renderer(sc)
    rng = np.random.RandomState(7)
    libm = C.CDLL("libm.so.6")
    libm.acosf.restype = C.c_float; libm.acosf.argtypes = [C.c_float]
    libm.atan2f.restype = C.c_float; libm.atan2f.argtypes = [C.c_float, C.c_float]
    x = np.concatenate([rng.uniform(-1, 1, 200000), np.array([1.0, -1.0, 0.0, 0.5, -0.5, 1e-9, 1.0000001, np.nan])]).astype(np.float32)
    dev = r.debug_libm(0, x)
    host = np.array([libm.acosf(float(v)) for v in x], np.float32)
    assert np.array_equal(dev.view(np.uint32)[~np.isnan(host)], host.view(np.uint32)[~np.isnan(host)]) and np.all(np.isnan(dev[np.isnan(host)]))
    y = rng.uniform(-2, 2, 200000).astype(np.float32); xx = rng.uniform(-2, 2, 200000).astype(np.float32)
    dev = r.debug_libm(1, y, xx)
    host = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y, xx)], np.float32)
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))
    # expf / log2f: glibc's own table-driven algorithms incl. its fused multiply-adds -> the host's bits, specials included
    libm.expf.restype = C.c_float; libm.expf.argtypes = [C.c_float]
    libm.log2f.restype = C.c_float; libm.log2f.argtypes = [C.c_float]
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.7, 88.8, -103.2, -103.5, -104.0, -200.0, 1e-45, 1e-40, 1.0, -1.0, 3.4e38], np.float32)
    v = np.concatenate([rng.uniform(-110, 90, 200000).astype(np.float32), (rng.uniform(-1, 1, 100000) * 1e-3).astype(np.float32),
                        rng.randint(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32), special])
    for fn_id, fn in ((2, libm.expf), (3, libm.log2f)):
        d = r.debug_libm(fn_id, v); h = np.array([fn(float(a)) for a in v], np.float32)
        ok = (d.view(np.uint32) == h.view(np.uint32)) | (np.isnan(d) & np.isnan(h))
        assert ok.all(), (fn_id, v[~ok][:5], d[~ok][:5], h[~ok][:5])
    # cvtss2si emulation incl. the out-of-range "indefinite" value
    v = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 3e9, -3e9, np.nan, 2147483520.0, 1e-3], np.float32)
    assert r.debug_libm(5, v).tolist()[:7] == [0.0, 2.0, 2.0, -0.0, -2.0, -2147483648.0, -2147483648.0]
