#!/usr/bin/env python3
"""Latency of ONE cfg3 frame rendered as K interleaved tile shards on K contexts at once (all writing the same framebuffer), against
one context rendering it alone.  K contexts x {one stream each (serial), two streams each}."""
import os, sys, time
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))
import torch
from pyrtx import api, host
sc = host.atrium_scene(1920, 1080, 3)
W, H, T = sc.width, sc.height, sc.tile_count
fb_rgb = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda"); fb_packed = torch.zeros(H * W, dtype=torch.int32, device="cuda")
KMAX = 4
rs = []
for k in range(KMAX):
    r = api.Renderer(sc); r.bind_framebuffer(fb_rgb.data_ptr(), fb_packed.data_ptr())
    st = torch.cuda.Stream(); r.set_stream(st.cuda_stream); rs.append((r, st))
def frame(K, serial):
    for k in range(K):
        rs[k][0].render_async(k, K, api.group_layout(W, H, K, k)[2], serial=serial)
def timed(K, serial, n=30):
    for _ in range(3):
        frame(K, serial); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        frame(K, serial); torch.cuda.synchronize()          # one frame at a time: wait for it before the next
    return 1000.0 * (time.perf_counter() - a) / n
ref = None
for K in (1, 2, 3, 4):
    for serial in (True, False):
        ms = timed(K, serial)
        if K == 1 and not serial:
            frame(1, False); torch.cuda.synchronize(); ref = fb_packed.clone()
        frame(K, serial); torch.cuda.synchronize()
        same = True if ref is None else bool(torch.equal(ref, fb_packed))
        print(f"K={K} contexts, {'one stream' if serial else 'two streams'} each: {ms:.3f} ms per frame (host-synchronised after every frame), frame identical: {same}")
