"""Shared helpers for the tests (golden loading, comparisons)."""
import gzip
import os

import numpy as np

from pyrtx import scene_io as sio

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(REPO, "tests", "golden")

# name -> (scene dir, golden file tag, config overrides)
GOLDENS = {
    "cube": ("cube", None, {}),
    "monkey": ("monkey", None, {}),
    "monkey_small": ("monkey_small", None, {}),
    "materials_aniso": ("materials", "aniso", {"mip_filter": 1, "texture_mode": 2}),
    "materials_trilinear": ("materials", "trilinear", {"mip_filter": 0, "texture_mode": 2}),
    "materials_ewa": ("materials", "ewa", {"mip_filter": 2, "texture_mode": 2}),
    "materials_bilinear": ("materials", "bilinear", {"mip_filter": 1, "texture_mode": 1}),
    # BVH_VISUALIZE_HEATMAP builds of the reference (Config.h:23, Raytracer.cpp:97-102) on the same two scenes
    "monkey_small_heat": ("monkey_small", "heat", {"heatmap": 1}),
    "materials_heat": ("materials", "heat", {"heatmap": 1, "mip_filter": 1, "texture_mode": 2}),
    # the remaining Config.h values, each from its own build of the reference (oracle/ref_harness/Makefile TRAVERSAL / TEXMODE / ANISO / BOUNCES):
    # BVH_TRAVERSE_TREE_NAIVE (Config.h:27), TEXTURE_SAMPLE_MODE_NEAREST (:38), MAX_ANISOTROPY 2 (:55), NUMBER_OF_BOUNCES 1 / 5 (:12)
    "materials_naive": ("materials", "naive", {"traversal_strategy": 0, "mip_filter": 1, "texture_mode": 2}),
    "materials_nearest": ("materials", "nearest", {"texture_mode": 0, "mip_filter": 1}),
    "materials_aniso2": ("materials", "aniso2", {"max_anisotropy": 2.0, "mip_filter": 1, "texture_mode": 2}),
    "materials_b1": ("materials", "b1", {"bounces": 1, "mip_filter": 1, "texture_mode": 2}),
    "materials_b5": ("materials", "b5", {"bounces": 5, "mip_filter": 1, "texture_mode": 2}),
    "camera_keys": ("camera_keys", None, {}),
    "dynamic": ("dynamic", None, {}),
    "tori16": ("tori16", None, {}),
    "tori16_f1": ("tori16_f1", None, {}),
}


def load_golden(name):
    d, tag, over = GOLDENS[name]
    with gzip.open(os.path.join(GOLDEN, d, "scene.rtxs.gz"), "rb") as f:
        sc = sio.load_scene(f.read())
    for k, v in over.items():
        sc.config[k] = v
    if over.get("texture_mode", 2) != 2:
        # non-mipmap builds of the reference never build mip chains (Texture.cpp:49-55)
        for t in sc.textures:
            t.desc["mipmapped"] = 0
            t.desc["mip_levels"] = 1
    g = dict(np.load(os.path.join(GOLDEN, d, f"golden{'_' + tag if tag else ''}.npz")))
    return sc, g


def compare_to_golden(out, g):
    """out: dict(rgb (H,W,3), packed (H,W), stats dict).  Returns dict of comparison figures."""
    if "xs" in g:
        rgb = out["rgb"][g["ys"], g["xs"]]
        packed = out["packed"][g["ys"], g["xs"]]
    else:
        rgb, packed = out["rgb"], out["packed"]
    diff = np.abs(rgb.astype(np.float64) - g["rgb"].astype(np.float64))
    both_nan = np.isnan(rgb) & np.isnan(g["rgb"])
    diff[both_nan] = 0.0
    return {
        "max_abs": float(np.nanmax(diff)) if diff.size else 0.0,
        "nan_mismatch": int((np.isnan(rgb) != np.isnan(g["rgb"])).sum()),
        "bit_exact": bool(np.array_equal(rgb.view(np.uint32), g["rgb"].view(np.uint32))),
        "n_diff_pixels": int((diff.max(axis=-1) > 0).sum()),
        "packed_mismatch": int((packed != g["packed"]).sum()),
        "stats_equal": [int(out["stats"][k]) for k in ("primary", "shadow", "reflection", "refraction")] == g["stats"].tolist(),
    }


def bit_exact(a, b):
    """fp32 arrays equal bit for bit; two NaNs count as equal whatever their sign / payload (x86 and gfx950 generate different default NaNs)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def load_fullsize_golden(name):
    """8 192 seeded pixels + stats + channel sums of a 1920x1080 frame rendered by the REAL reference
    (oracle/ref_harness/make_fullsize_goldens.py): atrium_1080p, tori16_1080p_f1..f3."""
    return dict(np.load(os.path.join(GOLDEN, "fullsize", name + ".npz")))


def compare_to_fullsize_golden(out, g):
    """Bit-exact on the sampled pixels, exact on the ray counts, exact on the whole-frame channel sums (float64 sums of fp32 values: equal
    iff the multiset of pixel values is — a checksum over all 2 073 600 pixels)."""
    rgb = out["rgb"][g["ys"], g["xs"]]
    sums = out["rgb"].astype(np.float64).sum(axis=(0, 1))
    return {"samples_bit_exact": bit_exact(rgb, g["rgb"]), "max_abs": float(np.nanmax(np.abs(rgb.astype(np.float64) - g["rgb"].astype(np.float64)))),
            "packed_equal": bool(np.array_equal(out["packed"][g["ys"], g["xs"]], g["packed"])),
            "stats_equal": [int(out["stats"][k]) for k in ("primary", "shadow", "reflection", "refraction")] == g["stats"].tolist(),
            "channel_sums_equal": bool(np.array_equal(sums, g["channel_sums"]))}
