"""the step after the path — Window::draw_quad (Window.cpp:87-95) through Data/Shaders/vertex.glsl and
fragment_fxaa.glsl / fragment_identity.glsl, as one HIP kernel (`rtx_present`).

PARITY UNPINNED against the reference: its shaders need an OpenGL driver and there is none in this environment (and no golden
screenshots of known scenes), so nothing here compares with reference output.  What is tested: the oracle's restatement of the
shaders has the properties the GLSL text implies, and the HIP kernel equals that restatement bit for bit."""
import numpy as np
import pytest

import util


def luma(img):
    import orc
    lut = orc.gamma_lut()
    return (np.float32(0.299) * lut[(img >> 16) & 255] + np.float32(0.587) * lut[(img >> 8) & 255]) + np.float32(0.114) * lut[img & 255]


def test_identity_is_the_gamma_curve_per_channel():
    import orc
    ramp = (np.arange(256, dtype=np.uint32)[None, :] * 0x010101).repeat(3, axis=0)
    out = orc.present(ramp, fxaa=False)
    lut = orc.gamma_lut()
    want = np.rint(np.clip(lut, 0, 1) * np.float32(255)).astype(np.uint32)        # round-to-nearest-even, as lrintf
    assert np.array_equal(out[1] & 255, want) and np.array_equal(out >> 16, out & 255)
    assert want[0] == 0 and want[255] == 255 and want[128] == 186               # 0.5^(1/2.2) = 0.7297 -> 186


def test_fxaa_leaves_flat_regions_alone_and_stays_inside_the_local_luma_range():
    import orc
    flat = np.full((24, 40), 0x4080c0, np.uint32)
    assert np.array_equal(orc.present(flat, True), orc.present(flat, False))
    rng = np.random.default_rng(7)
    img = np.full((48, 64), 0x202020, np.uint32)
    img[:, 32:] = 0xe0e0e0                                                        # a vertical edge
    img[10:20, 5:25] = rng.integers(0, 1 << 24, (10, 20), dtype=np.uint32)        # and some noise
    a, b = orc.present(img, True), orc.present(img, False)
    assert (a != b).any() and np.array_equal(a[30:, :28], b[30:, :28])            # only pixels near contrast change
    # grey edge columns: blended values lie between the two sides
    edge = a[40, 28:36] & 255
    assert edge.min() >= (b[40, 0] & 255) and edge.max() <= (b[40, 63] & 255)


def test_taps_wrap_around_the_frame_like_gl_repeat():
    """GL_REPEAT addressing makes the whole filter commute with cyclic shifts of the frame."""
    import orc
    rng = np.random.default_rng(11)
    img = rng.integers(0, 1 << 24, (40, 56), dtype=np.uint32)
    img[8:30, 10:40] = 0x303030                                                   # a flat patch so that long blur spans occur at its border
    base = orc.present(img, True)
    assert (base != orc.present(img, False)).mean() > 0.05
    for dy, dx in ((1, 0), (0, 1), (7, 13), (39, 55)):
        shifted = np.roll(img, (dy, dx), axis=(0, 1))
        assert np.array_equal(orc.present(shifted, True), np.roll(base, (dy, dx), axis=(0, 1))), (dy, dx)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cube", "materials_aniso", "monkey_small"])
def test_present_kernel_equals_the_oracle(name):
    import orc
    from pyrtx import api
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    out = r.render()
    assert np.array_equal(out["packed"], g["packed"])
    for fxaa in (False, True):
        assert np.array_equal(r.present(fxaa), orc.present(out["packed"], fxaa)), fxaa


@pytest.mark.gpu
def test_present_full_hd_and_odd_sizes():
    import orc
    from pyrtx import api, host
    sc = host.atrium_scene(1920, 1080, 3, detail=0)
    r = api.Renderer(sc)
    out = r.render()
    assert np.array_equal(r.present(True), orc.present(out["packed"], True))
    sc2 = host.atrium_scene(333, 207, 1, detail=0)                                # not multiples of the 16x16 blocks / 32x32 tiles
    r2 = api.Renderer(sc2)
    out2 = r2.render()
    assert np.array_equal(r2.present(True), orc.present(out2["packed"], True))
    assert np.array_equal(r2.present(False), orc.present(out2["packed"], False))
