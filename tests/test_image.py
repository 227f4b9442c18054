"""texture files -> texels (Texture::load, Texture.cpp:30-129) in librtx_host.so.

Pinned by the reference's own vendored decoder: tests/golden/images/expected.npz holds what stb_image v2.19
(stbi_load(..., STBI_rgb_alpha), Texture.cpp:40; built from /root/reference/include by oracle/ref_harness) returned for
each fixture file; oracle/ref_harness/make_image_goldens.py additionally compared all 51 .png/.tga files of the
reference's Data/ directory (reference_data_check.json).  floor.png and heat_palette.png are data files of the reference."""
import json
import os

import numpy as np
import pytest

import util
from pyrtx import host, scene_io as sio

D = os.path.join(util.GOLDEN, "images")
EXPECTED = np.load(os.path.join(D, "expected.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_decoder_matches_stb_image(name):
    got = host.load_image(os.path.join(D, name))
    assert got.shape == EXPECTED[name].shape
    assert np.array_equal(got, EXPECTED[name])


def test_fixture_set_covers_the_formats():
    names = set(EXPECTED.files)
    for color, depths in {0: (1, 2, 4, 8, 16), 2: (8, 16), 3: (1, 2, 4, 8), 4: (8, 16), 6: (8, 16)}.items():
        for d in depths:
            assert any(n.startswith(f"png_c{color}_d{d}_adam7") for n in names) and any(n.startswith(f"png_c{color}_d{d}_plain") for n in names)
    for t, b in [(1, 8), (2, 16), (2, 24), (2, 32), (3, 8), (3, 16)]:
        assert any(n.startswith(f"tga_t{t}_b{b}") and "_rle" in n for n in names) and any(n.startswith(f"tga_t{t}_b{b}") and "_rle" not in n for n in names)
    rep = json.load(open(os.path.join(D, "reference_data_check.json")))
    assert rep["identical"] == 51 and rep["files"] == 52          # the one JPEG (not referenced by any .mtl) is refused, see below


def test_texture_load_equals_the_texture_the_reference_sampled():
    """floor.png through rtxh_texture_load == the texel array dumped from the reference's Texture after Texture::load
    (golden `dynamic` scene: 32x32, sRGB decode with the host powf, 6 box-filtered levels)."""
    sc, _ = util.load_golden("dynamic")
    ref = sc.textures[0]
    t = host.load_texture(os.path.join(D, "floor.png"))
    assert t.desc.tobytes() == ref.desc.tobytes()
    assert t.texels.tobytes() == ref.texels.tobytes()
    flat = host.load_texture(os.path.join(D, "floor.png"), mipmap_mode=False)          # TEXTURE_SAMPLE_MODE != MIPMAP
    assert int(flat.desc["mip_levels"][0]) == 1 and int(flat.desc["mipmapped"][0]) == 0
    assert flat.texels.tobytes() == ref.texels[:32 * 32].tobytes()
    odd = host.load_texture(os.path.join(D, "heat_palette.png"))                        # 501x1: not a power of two -> no mips
    assert (int(odd.desc["width"][0]), int(odd.desc["height"][0]), int(odd.desc["mipmapped"][0]), len(odd.texels)) == (501, 1, 0, 501)
    assert np.array_equal(odd.texels, host.srgb8_to_linear(EXPECTED["heat_palette.png"][0, :, :3]))


def test_decoder_errors(tmp_path):
    def status(data):
        p = tmp_path / "x.bin"
        p.write_bytes(data)
        with pytest.raises(ValueError) as e:
            host.load_image(str(p))
        return int(str(e.value).split("status ")[-1])
    with pytest.raises(ValueError) as e:
        host.load_image(str(tmp_path / "missing.png"))
    assert str(e.value).endswith("status 5")                                            # RTX_ERR_STATE: unreadable (the reference aborts)
    assert status(b"\xff\xd8\xff\xe0" + bytes(64)) == 4                                 # JPEG: known to stb_image, not to this loader
    assert status(b"BM" + bytes(64)) == 4
    assert status(bytes([0, 7, 2]) + bytes(64)) == 1                                    # neither PNG nor an acceptable TGA header
    png = open(os.path.join(D, "png_c2_d8_plain_trns.png"), "rb").read()
    assert status(png[:len(png) // 2]) == 1                                             # truncated
    broken = bytearray(png); broken[8 + 8 + 8] = 3                                      # bit depth 3
    assert status(bytes(broken)) == 1
    assert status(b"") == 1


def test_png_writer_round_trips_through_the_decoder(tmp_path):
    rng = np.random.default_rng(3)
    for h, w in ((1, 1), (7, 300), (200, 333)):                                          # the last one needs several stored deflate blocks
        frame = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
        p = str(tmp_path / f"f{h}x{w}.png")
        host.save_png(p, frame)
        back = host.load_image(p)
        assert back.shape == (h, w, 4) and np.all(back[..., 3] == 255)
        assert np.array_equal((back[..., 0].astype(np.uint32) << 16) | (back[..., 1].astype(np.uint32) << 8) | back[..., 2], frame)
        import zlib, struct
        raw = open(p, "rb").read()                                                       # and a standard zlib accepts the stream + CRCs
        pos, idat = 8, b""
        while pos < len(raw):
            n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
            assert zlib.crc32(raw[pos + 4:pos + 8 + n]) & 0xffffffff == struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0]
            if tag == b"IDAT":
                idat += raw[pos + 8:pos + 8 + n]
            pos += 12 + n
        assert len(zlib.decompress(idat)) == h * (3 * w + 1)


def test_decoder_survives_mutated_files(tmp_path):
    """Texture files are untrusted input: bit flips, truncation, inserted bytes and 0xffffffff fields must decode or be rejected with
    a status — never crash, hang or allocate what the header merely claims (the same loop ran clean under ASAN + UBSAN)."""
    import time
    rng = np.random.default_rng(5)
    files = [f for f in sorted(os.listdir(D)) if f.endswith((".png", ".tga"))]
    decoded = rejected = 0
    t0 = time.time()
    for f in files:
        raw = open(os.path.join(D, f), "rb").read()
        for k in range(12):
            b = bytearray(raw)
            mode = k % 4
            if mode == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif mode == 1:
                b = b[:int(rng.integers(0, len(b)))]
            elif mode == 2:
                i = int(rng.integers(0, len(b))); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40))).astype(np.uint8))
            else:
                i = int(rng.integers(0, max(1, len(b) - 8))); b[i:i + 4] = b"\xff\xff\xff\xff"
            p = tmp_path / "m.bin"
            p.write_bytes(bytes(b))
            try:
                img = host.load_image(str(p)); decoded += 1
                assert img.ndim == 3 and img.shape[2] == 4
            except ValueError:
                rejected += 1
    assert decoded > 100 and rejected > 100 and time.time() - t0 < 60


def test_decoder_rejects_size_bombs_quickly(tmp_path):
    import struct, time, zlib
    t0 = time.time()
    tga = struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, 65535, 32000, 24, 0) + b"\x00" * 16          # 2 G pixels claimed, 16 bytes of data
    p = tmp_path / "bomb.tga"; p.write_bytes(tga)
    with pytest.raises(ValueError):
        host.load_image(str(p))
    def chunk(tag, d): return struct.pack(">I", len(d)) + tag + d + struct.pack(">I", zlib.crc32(tag + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1 << 24, 60, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\x00" * 1000)) + chunk(b"IEND", b"")
    p = tmp_path / "bomb.png"; p.write_bytes(png)
    with pytest.raises(ValueError):
        host.load_image(str(p))
    ratio = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\x00" * (64 << 20), 9)) + chunk(b"IEND", b"")
    p = tmp_path / "ratio.png"; p.write_bytes(ratio)                                                          # 64 MB of zeros behind a 4x4 header
    assert host.load_image(str(p)).shape == (4, 4, 4)                                                          # decodes; the surplus is never inflated
    assert time.time() - t0 < 10
