"""Writes a triangle soup as Wavefront OBJ / MTL (+ uncompressed TGA textures) in the form the reference's loader
reads back (OBJLoader.cpp:8-187 over tinyobjloader; Texture::load, Texture.cpp:30-129): one `v/vt/vn` triple per
corner, faces grouped by `usemtl`, texture v flipped because the loader flips it again (OBJLoader.cpp:139-141).
Used to hand the procedural cfg3 mesh to the real reference (oracle/ref_harness/atrium_crosscheck.py) and by
tests/test_sbvh.py, which runs the same text through this repo's own loader."""
import os

import numpy as np


def write_tga(path: str, img_u8: np.ndarray) -> None:
    h, w, _ = img_u8.shape
    hdr = bytearray(18)
    hdr[2] = 2; hdr[12] = w & 255; hdr[13] = w >> 8; hdr[14] = h & 255; hdr[15] = h >> 8; hdr[16] = 24; hdr[17] = 0x20   # top-left origin
    with open(path, "wb") as f:
        f.write(bytes(hdr)); f.write(img_u8[:, :, ::-1].tobytes())


def write_obj(directory: str, name: str, pos, nrm, uv, mid, mats, texture_images=()) -> str:
    """pos/nrm (n,3,3), uv (n,3,2) in the reference's in-memory convention, mid (n,) mesh-local material ids,
    mats: MATERIAL records (texture_id indexes texture_images).  Returns the path of the .obj."""
    os.makedirs(directory, exist_ok=True)
    for i, img in enumerate(texture_images):
        write_tga(os.path.join(directory, f"tex{i}.tga"), img)
    with open(os.path.join(directory, name + ".mtl"), "w") as f:
        for i, m in enumerate(mats):
            f.write(f"newmtl m{i}\nKd {m['diffuse'][0]:.9g} {m['diffuse'][1]:.9g} {m['diffuse'][2]:.9g}\n"
                    f"Ks {m['reflection'][0]:.9g} {m['reflection'][1]:.9g} {m['reflection'][2]:.9g}\n"
                    f"Kt {m['transmittance'][0]:.9g} {m['transmittance'][1]:.9g} {m['transmittance'][2]:.9g}\nNi {m['index_of_refraction']:.9g}\n")
            if m["texture_id"] >= 0:
                f.write(f"map_Kd tex{int(m['texture_id'])}.tga\n")
            f.write("\n")
    path = os.path.join(directory, name + ".obj")
    with open(path, "w") as f:
        f.write(f"mtllib {name}.mtl\n")
        p = np.asarray(pos).reshape(-1, 3); nn = np.asarray(nrm).reshape(-1, 3); t = np.asarray(uv).reshape(-1, 2).copy()
        t[:, 1] = 1.0 - t[:, 1]
        f.write("".join(f"v {a:.9g} {b:.9g} {c:.9g}\n" for a, b, c in p))
        f.write("".join(f"vt {a:.9g} {b:.9g}\n" for a, b in t))
        f.write("".join(f"vn {a:.9g} {b:.9g} {c:.9g}\n" for a, b, c in nn))
        mid = np.asarray(mid)
        cur = -1
        out = []
        for k in np.argsort(mid, kind="stable"):
            if mid[k] != cur:
                cur = int(mid[k]); out.append(f"usemtl m{cur}\n")
            a = 3 * k + 1
            out.append(f"f {a}/{a}/{a} {a+1}/{a+1}/{a+1} {a+2}/{a+2}/{a+2}\n")
        f.write("".join(out))
    return path
