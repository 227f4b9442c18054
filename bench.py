#!/usr/bin/env python3
"""bench.py — headline benchmark of the render path on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow + reflection + refraction, counted exactly as the
reference's PerformanceStats at SIMD_LANE_SIZE 1) and ms/frame, "Sponza 1920x1080" at 1/2/4/8 GPUs.
Sponza itself is absent from the reference mount, so the workload is the seeded Sponza-class
stand-in of SURVEY.md 8(d) ("atrium": 255 296 triangles, 25 materials, 3 lights, depth 3,
anisotropic mip filter).

A "step" is one full frame: rtx_render_tiles over this rank's tiles (inputs resident in HBM) and,
for N > 1, the RCCL gather of the packed framebuffer tiles to rank 0 (BASELINE configs[3]); the frame
is fixed as N grows, so scaling is "strong".  Launch: `python bench.py --gpus N --steps K --warmup W` — for N > 1 that process
starts the N ranks itself (one fresh process per GPU, before anything touches the GPU; `--dry-launch` prints them) — or under
torch.distributed.run, which sets WORLD_SIZE and makes bench.py one of the ranks.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))

import numpy as np  # noqa: E402


def algorithmic_bytes_closest(work, level0_rays):
    """Algorithmic bytes moved by all k_trace<closest> launches of one frame:
    32 B per BVH node popped (BVHNode.h:10), 64 B inverse matrix per instance entered (Mesh.h:10),
    36 B TriangleHot per triangle test (BottomLevelBVH.h:6-10), 24 B origin+direction read per queued
    ray (primary rays are generated, not read), 20 B hit record written per ray."""
    rays = work["closest_rays"]
    return (32 * (work["tlas_nodes_closest"] + work["blas_nodes_closest"]) + 64 * work["instances_closest"]
            + 36 * work["tri_tests_closest"] + 24 * max(rays - level0_rays, 0) + 20 * rays)


def algorithmic_bytes_any(work):
    """k_trace<any>: node / instance / triangle bytes as above, 28 B shadow ray read (origin, direction,
    max distance) and 4 B occlusion flag written per shadow ray."""
    return (32 * (work["tlas_nodes_any"] + work["blas_nodes_any"]) + 64 * work["instances_any"]
            + 36 * work["tri_tests_any"] + 32 * work["any_rays"])


def slab_equivalents(work, kind, tri_cost):
    """The traversal ARITHMETIC of one frame's closest-hit / shadow-ray kernels in units of one AABB::intersect: the reference's own
    call counts at SIMD_LANE_SIZE 1 (AABB::intersect per node popped, triangle_trace / triangle_intersect per triangle tested; counted
    by the instrumented kernels, equal to the oracle's counters) with a triangle test priced at `tri_cost` box tests (ratio of the two
    rates experiments/slab_rate.hip measures)."""
    sfx = "closest" if kind == "closest" else "any"
    return work["tlas_nodes_" + sfx] + work["blas_nodes_" + sfx] + tri_cost * work["tri_tests_" + sfx]


def roofline_peak():
    """The chip's rate for the traversal arithmetic alone, operands in registers (experiments/slab_rate.hip): measured live when the
    binary is here (built by __graft_entry__.build()), else the committed profiles/ figure.  Returns (G slab tests/s, G triangle tests/s, source)."""
    import subprocess
    exe = os.path.join(REPO, "experiments", "slab_rate")
    rows, src = [], None
    try:
        if os.path.exists(exe):
            out = subprocess.run([exe], capture_output=True, text=True, timeout=120, check=True).stdout
            rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
            src = "experiments/slab_rate run inside this bench.py invocation"
    except Exception as e:
        sys.stderr.write(f"live roofline peak skipped: {e!r}\n")
    if not rows:
        import glob
        files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_roofline_peak.jsonl")))
        if files:
            rows = [json.loads(l) for l in open(files[-1]) if l.startswith("{")]
            src = os.path.relpath(files[-1], REPO)
    if not rows:
        return None
    best = max(rows, key=lambda r: r["slab_tests_G_per_s"])
    return best["slab_tests_G_per_s"], max(r["triangle_tests_G_per_s"] for r in rows), src


def _group(times):
    g = {}
    for name, ms in times:
        g.setdefault(name, []).append(ms)
    return g


def cpu_quota():
    """CPU share the container actually grants (cgroup v2 cpu.max / v1 cfs quota), in cores, or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q <= 0 else round(q / per, 2)
        except Exception:
            return None


def baseline_threads():
    """Threads for the CPU baselines: what the container may actually run in parallel (its CPU quota), not the host's core count."""
    n = os.cpu_count() or 1
    q = cpu_quota()
    return max(1, min(n, int(q + 0.999))) if q else n


def reference_cpu_baseline(args, rays_per_frame, stats):
    """The REAL reference (oracle/_ref/refdump_*, built from /root/reference in the build container; only the binary travels) on this
    box's host cores: its own tile loop on a thread pool (WorkerThread.cpp:53-65), SIMD_LANE_SIZE 1, on the same mesh / lights / camera
    (handed over as OBJ + MTL + TGA text, the way the reference loads scenes).  Returns None when the binary is absent or anything fails."""
    import shutil, subprocess, tempfile
    exe = os.path.join(REPO, "oracle", "_ref", "refdump_s0_m1_b3_t2")
    if args.workload != "atrium" or not os.path.exists(exe):
        return None
    work = tempfile.mkdtemp(prefix="rtx_refbase_")
    try:
        from pyrtx import host, obj_export
        data = os.path.join(work, "Data")
        os.makedirs(os.path.join(data, "Sky_Probes")); os.makedirs(os.path.join(data, "sponza"))
        host.synthetic_sky().tofile(os.path.join(data, "Sky_Probes", "rnl_probe.float"))
        meshes = os.path.join(REPO, "tests", "golden", "meshes")              # SCENE_SPONZA's constructor opens three meshes before the script
        for dst in ("sponza/sponza.obj", "Magnifier.obj", "Concave.obj"):     # replaces them (Scene.cpp:77-79): stage a small one under each name
            shutil.copy(os.path.join(meshes, "Diamond.obj"), os.path.join(data, dst))
        shutil.copy(os.path.join(meshes, "Diamond.mtl"), os.path.join(data, "sponza", "Diamond.mtl"))
        shutil.copy(os.path.join(meshes, "Diamond.mtl"), os.path.join(data, "Diamond.mtl"))
        seed = 0x5EED0003
        pos, nrm, uv, mid, nmat = host.atrium_mesh(seed, args.detail)
        obj_export.write_obj(os.path.join(data, "atrium"), "atrium", pos, nrm, uv, mid, host.atrium_materials(nmat), host.procedural_texture_images(seed))
        cam, pt, sp, dr = host.ATRIUM_CAMERA, host.ATRIUM_POINT, host.ATRIUM_SPOT, host.ATRIUM_DIR
        with open(os.path.join(work, "scene.txt"), "w") as f:
            f.write(f"size {args.width} {args.height}\nbounces 3\nmesh ./Data/atrium/atrium.obj 0 0 0\n")
            f.write("point %g %g %g %g %g %g\n" % (*pt[0], *pt[1]))
            f.write("spot %g %g %g %g %g %g %.9g %.9g %.9g %g %g\n" % (*sp[0], *sp[1], *sp[2], sp[3], sp[4]))
            f.write("dir %g %g %g %.9g %.9g %.9g\n" % (*dr[0], *dr[1]))
            f.write("camera_axis_angle %g %g %g %g %g %g %.9g\n" % (*cam[0], *cam[1][0], cam[1][1]))
        cores = baseline_threads()
        out = os.path.join(work, "out")
        subprocess.run([exe, "scene.txt", out, "--threads", str(cores), "--frames", "5", "--timing-only"], cwd=work, check=True, timeout=600,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t = json.load(open(os.path.join(out, "timing.json")))
        ref_stats = {"primary": t["primary"], "shadow": t["shadow"], "reflection": t["reflection"], "refraction": t["refraction"]}
        ms = t["render_ms_threads_best"]
        return {"value": round(sum(ref_stats.values()) / (ms * 1e-3) * 1e-6, 3), "unit": "Mrays/s", "cores": cores, "kind": "reference",
                "sample": f"best of 5 full frames of the same workload by the real reference (clayne/CPU-Raytracer sources compiled with g++ -O2, SIMD_LANE_SIZE 1, its own SBVH), "
                          f"its tile loop on {cores} threads; one thread: {t['render_tile_ms_1thread']:.0f} ms per frame",
                "ms_per_frame": round(ms, 2), "ray_counts_equal_to_gpu": ref_stats == stats, "container_cpu_quota_cores": cpu_quota()}
    except Exception as e:                                                  # never let the baseline break the benchmark line
        sys.stderr.write(f"reference baseline skipped: {e!r}\n")
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def shade_bytes(work, stats, lights):
    """Algorithmic bytes of one frame's k_shade launches: what Raytracer::bounce reads and writes between the trace and the
    recursion, at the sizes of this repo's queue records.  Reads: hit record 20 B per ray, the ray record 80 B per secondary ray (primary rays
    are regenerated), TriangleCold 64 B + two edges 32 B per triangle hit (the instance's matrices: a handful of lines, L1-resident), Material 56 B per shaded hit,
    16 B per texel fetched, 12 B per sky lookup.  Writes: node record 32 B per ray + 32 B per hit, point + normal 32 B per lit hit (the shadow rays
    themselves are derived by their consumers), 80 B per spawned ray."""
    rays = work["closest_rays"]; secondary = rays - stats["primary"]
    lit_hits = stats["shadow"] // max(lights, 1)
    reads = 20 * rays + 80 * secondary + (64 + 32) * work["triangle_hits"] + 56 * work["shaded_hits"] + 16 * work["texel_fetches"] + 12 * work["sky_lookups"]
    writes = 32 * rays + 32 * work["shaded_hits"] + 32 * lit_hits + 80 * work["rays_spawned"]
    return reads + writes


def resolve_bytes(work, stats, lights):
    """k_resolve: node record 32 B per ray + 32 B per hit read, point + normal 32 B per lit hit, occlusion flag 4 B per shadow ray, a child's colour
    16 B read and written per spawned ray, 16 B of framebuffer (fp32 RGB + packed) written per primary ray."""
    lit_hits = stats["shadow"] // max(lights, 1)
    return 32 * work["closest_rays"] + 32 * work["shaded_hits"] + 32 * lit_hits + 4 * stats["shadow"] + 32 * work["rays_spawned"] + 16 * stats["primary"]


def sources_sha16():
    """Fingerprint of the kernel sources: profiles/*_hbm_traffic.json carries the one it was measured on (tools/hbm_traffic.sh)."""
    import hashlib, glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "cpu-raytracer_amd", "csrc", "*.h")) + glob.glob(os.path.join(REPO, "cpu-raytracer_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_roofline(args, sc, work, stats, level0_rays, kern, alone, ms_per_step, tot):
    """ONE statement of where the frame stands against the chip:
      headline   whole-frame traversal arithmetic (the reference's own AABB::intersect + triangle-test calls for this frame's rays, triangle tests
                 priced in box tests) / ms_per_step of the timed region / the chip's rate for that arithmetic alone (experiments/slab_rate, run here);
      kernel     the dominant kernel: the one whose launches need the most time per frame with the chip to themselves (serial-mode pass; without
                 that pass: the most GPU time per step in the timed region, HIP events on its launch stream);
      per_kernel every frame kernel against ITS bound on ONE denominator: its launches' time per frame with nothing else on the chip
                 (the serial-mode pass after the timed region) — traversal kernels against the VALU ceiling (closest hit also against the
                 divergent-gather ceiling), k_shade / k_resolve against HBM bandwidth;
      traffic    PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) per frame from the newest profiles/r*_hbm_traffic.json, with whether it was measured on
                 the kernel sources being run; hbm = the algorithmic-bytes view of SURVEY.md 8(d), which does not bind this design."""
    import glob
    peak = None if args.no_extras else roofline_peak()
    name_of = {"k_trace_closest_shade": "k_trace_closest"}
    per_step = {}
    for k, v in kern.items():
        per_step[name_of.get(k, k)] = per_step.get(name_of.get(k, k), 0.0) + v["total_ms_per_step"]
    # dominant kernel: by the time its launches need with the chip to themselves where that pass ran (the timed region's per-kernel GPU time also counts
    # the time a launch's waves wait for room beside the other frames' persistent grids: k_shade's 0.21 ms become 0.9-1.0 there, level with closest hit)
    alone_named = {}
    for k, v in (alone or {}).items():          # the item launch belongs to the shadow-ray walk (per_kernel groups them the same way)
        if v: alone_named[{"k_trace_items": "k_trace_any"}.get(name_of.get(k, k), name_of.get(k, k))] = alone_named.get({"k_trace_items": "k_trace_any"}.get(name_of.get(k, k), name_of.get(k, k)), 0.0) + v
    dominant = max(alone_named, key=alone_named.get) if alone_named else (max(per_step, key=per_step.get) if per_step else None)
    traffic = traffic_src = tj = None
    tfiles = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")))
    if tfiles:
        try:
            tj = json.load(open(tfiles[-1])); traffic_src = os.path.relpath(tfiles[-1], REPO)
        except Exception:
            tj = None
    lpf = {k: v["launches_per_frame"] for k, v in kern.items()}
    def traffic_per_frame(k):
        if not tj or tj.get(k + "_bytes_per_launch") is None or k not in lpf:
            return None
        return int(tj[k + "_bytes_per_launch"] * lpf[k])
    roof = {"bound": "valu", "kernel": dominant, "achieved": None, "peak": None, "unit": "G AABB::intersect equivalents/s", "frac": None,
            "traffic": traffic_per_frame(dominant) if dominant else None,
            "traffic_note": None if not tj else f"PMC bytes per frame of {dominant}'s launches (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included), {traffic_src}",
            "traffic_measured_on_these_sources": None if not tj else (tj.get("sources_sha16") == sources_sha16()),
            "gpu_ms_per_step_by_kernel": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
            "definition": "achieved = (AABB::intersect calls + triangle tests x their cost in box tests, closest-hit and shadow rays of one frame, the reference's own counts at "
                          "SIMD_LANE_SIZE 1) / ms_per_step; peak = the chip's rate for that arithmetic with every operand in registers (experiments/slab_rate, 17 VALU per box test); "
                          "kernel = most time per frame ALONE on the chip (gpu_ms_per_step_by_kernel: GPU time in the timed region, where a launch also waits for room beside other frames' grids); "
                          "per_kernel fractions use each kernel's time per frame ALONE on the chip"}
    if not peak:
        return roof
    slab_peak, tri_peak, peak_src = peak
    tri_cost = slab_peak / tri_peak
    units = {"k_trace_closest": slab_equivalents(work, "closest", tri_cost), "k_trace_any": slab_equivalents(work, "any", tri_cost)}
    frame_units = units["k_trace_closest"] + units["k_trace_any"]
    ach = frame_units / (ms_per_step * 1e-3) * 1e-9
    roof.update({"achieved": round(ach, 2), "peak": round(slab_peak, 1), "frac": round(ach / slab_peak, 4), "units_per_frame": int(frame_units),
                 "triangle_test_cost_in_box_tests": round(tri_cost, 3), "peak_source": peak_src})
    pk = {}
    if alone:
        any_ms = alone.get("k_trace_any", 0.0) + alone.get("k_trace_items", 0.0)
        if alone.get("k_trace_closest"):
            t = alone["k_trace_closest"] * 1e-3
            pk["k_trace_closest"] = {"bound": "valu", "ms_per_frame_alone": round(alone["k_trace_closest"], 4), "units_per_frame": int(units["k_trace_closest"]),
                                     "frac": round(units["k_trace_closest"] / t * 1e-9 / slab_peak, 4)}
            gfiles = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_gather_occupancy.txt")))
            ceil = None
            if gfiles:
                for line in open(gfiles[-1]):
                    f = line.split()
                    if len(f) > 8 and f[0] == "table" and f[1] == "32" and f[4].rstrip(":") == "4":
                        ceil = float(f[7])
            if ceil:       # the per-lane walk's other bound: divergent 64-byte gathers through the vector memory path (experiments/gather_occupancy)
                recs = work["blas_nodes_closest"] / 2 + work["tri_tests_closest"] + work["instances_closest"]
                pk["k_trace_closest"]["gather"] = {"bound": "divergent 64-byte gathers", "records_per_frame": int(recs), "G_records_per_s": round(recs / t * 1e-9, 1),
                                                   "ceiling_G_records_per_s": ceil, "frac": round(recs / t * 1e-9 / ceil, 3), "ceiling_source": os.path.relpath(gfiles[-1], REPO),
                                                   "records": "ALGORITHMIC: the child pairs, triangles and instance entries of the reference's binary walk (instrumented pass); the kernel's private lanes fetch ordered 4-wide records, i.e. fewer and larger ones, so this fraction compares the work done with what a binary per-lane walk could reach and may exceed 1"}
        if any_ms:
            pk["k_trace_any"] = {"bound": "valu", "includes": "k_trace_items (the launch that walks the items the packet launch hands over)", "ms_per_frame_alone": round(any_ms, 4),
                                 "units_per_frame": int(units["k_trace_any"]), "frac": round(units["k_trace_any"] / (any_ms * 1e-3) * 1e-9 / slab_peak, 4)}
        n_lights = sc.point_lights.size + sc.spot_lights.size + sc.dir_lights.size
        for k, nbytes in (("k_shade", shade_bytes(work, stats, n_lights)), ("k_resolve", resolve_bytes(work, stats, n_lights))):
            if alone.get(k):
                tr = traffic_per_frame(k)
                pk[k] = {"bound": "hbm", "ms_per_frame_alone": round(alone[k], 4), "algorithmic_bytes_per_frame": int(nbytes),
                         "achieved_GBps": round(nbytes / (alone[k] * 1e-3) / 1e9, 1), "peak_GBps": 8000.0, "frac": round(nbytes / (alone[k] * 1e-3) / 8e12, 4),
                         "traffic_bytes_per_frame": tr, "traffic_over_algorithmic": None if not tr else round(tr / nbytes, 2),
                         "traffic_frac_of_peak": None if not tr else round(tr / (alone[k] * 1e-3) / 8e12, 4)}
        for k in pk:
            if k in per_step:
                pk[k]["gpu_ms_per_step_in_timed_region"] = round(per_step[k] + (per_step.get("k_trace_items", 0.0) if k == "k_trace_any" else 0.0), 4)
    roof["per_kernel"] = pk
    alg_bytes = algorithmic_bytes_closest(work, level0_rays) + algorithmic_bytes_any(work)
    frame_traffic = None if not tj else sum(t for t in (traffic_per_frame(k) for k in lpf) if t)
    roof["hbm"] = {"binding": False, "algorithmic_bytes_per_frame_traversal": int(alg_bytes), "algorithmic_GBps_at_ms_per_step": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                   "peak_GBps": 8000.0, "traffic_bytes_per_frame_all_kernels": frame_traffic,
                   "traffic_frac_of_peak": None if not frame_traffic else round(frame_traffic / (ms_per_step * 1e-3) / 8e12, 4),
                   "note": "SURVEY 8(d)'s byte model (reference struct sizes x nodes / triangles visited) exceeds the HBM peak at this frame rate without any work being "
                           "skipped: a packet fetches a node once for 64 rays and the 31 MB scene is cache-resident, so the measured traffic is a small fraction of it"}
    return roof



def launch_plan(n_gpus, argv, master_port=None, env=None):
    """`python bench.py --gpus N` without a launcher around it: the ranks this process will start, one fresh process per GPU (what
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` would start: same environment variables, rank r on device r).
    Pure function of its arguments: tests/test_bench_launch.py checks it without a GPU."""
    env = dict(os.environ if env is None else env)
    port = int(master_port or env.get("MASTER_PORT") or 29400 + (os.getpid() % 500))
    plan = []
    for r in range(n_gpus):
        e = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus), "LOCAL_WORLD_SIZE": str(n_gpus),
             "MASTER_ADDR": env.get("MASTER_ADDR", "127.0.0.1"), "MASTER_PORT": str(port),
             "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}      # dmabuf IPC: RCCL between processes needs it on this driver
        plan.append({"rank": r, "device": r, "env": e, "argv": [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]})
    return plan


def self_launch(args, argv):
    """Starts the N ranks as child processes BEFORE this process has touched the GPU (it never does: no torch import, no HIP call), relays
    rank 0's one JSON line, and exits with the first failing rank's code.  Children are ended by PID, never by pattern."""
    import subprocess
    plan = launch_plan(args.gpus, argv)
    if args.dry_launch:
        print(json.dumps({"launcher": "bench.py self-launch (one process per GPU, rank r on device r)", "n_gpus": args.gpus, "ranks": plan}))
        return 0
    procs = []
    for p in plan:
        env = dict(os.environ); env.update(p["env"])
        procs.append(subprocess.Popen(p["argv"], env=env, stdout=None if p["rank"] == 0 else subprocess.DEVNULL))      # rank 0 prints the line
    rc, deadline = 0, time.time() + 3000
    pending = list(procs)
    while pending:
        for pr in list(pending):
            code = pr.poll()
            if code is None:
                continue
            pending.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:          # a rank died: the others would wait for it in the next collective for ever
                    other.terminate()
        if time.time() > deadline:
            for other in pending:
                other.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no launcher around bench.py: print the ranks / devices / environment it would start, and exit")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="atrium", choices=["atrium", "monkey", "tori16", "materials"])
    ap.add_argument("--detail", type=int, default=1)
    ap.add_argument("--accel", choices=["sbvh", "bvh", "binned"], default="sbvh",
                    help="BLAS builder for the atrium: the reference's SBVH restated (default, BASELINE configs[2]), its plain BVH, or this repo's binned SAH")
    ap.add_argument("--bounces", type=int, default=3, help="NUMBER_OF_BOUNCES of the atrium workload (BASELINE configs[2]: 3); other values are A/B runs, not the headline")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--no-reference-baseline", action="store_true", help="CPU baseline from the oracle port only (skip the run of oracle/_ref's real reference)")
    ap.add_argument("--simple-trace", action="store_true", help="A/B: plain pop-and-test traversal kernels")
    ap.add_argument("--frames-in-flight", type=int, default=3, help="contexts rendering consecutive frames concurrently (1 = one frame at a time)")
    ap.add_argument("--simulate-ranks", type=int, default=0, help="single-GPU rehearsal: render only the tile shard rank 0 of N would own (no gather)")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL gather plumbing even with one rank (single-GPU rehearsal of the N>1 path)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra passes (serial-mode timings, culling figure): use under rocprofv3 so its per-kernel averages cover only the benchmark's own launches")
    ap.add_argument("--serial", action="store_true", help="RTX_RENDER_SERIAL for the timed loop (one stream per frame); default when more than one frame is in flight")
    ap.add_argument("--overlap", action="store_true", help="force the two-stream intra-frame overlap even with several frames in flight")
    ap.add_argument("--no-kernel-timing", action="store_true", help="A/B: no HIP events around the launches of the timed region (roofline fields become null)")
    ap.add_argument("--cull", action="store_true", help="time with RTX_RENDER_CULL_DEAD_SHADOW_RAYS as the main figure (default: reported as an extra)")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own brings up its N ranks itself; under torch.distributed.run (WORLD_SIZE set) it IS one of the ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.dry_launch:
        print(json.dumps({"launcher": "none needed", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "ranks": []}))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    from pyrtx import api, host

    if world != max(args.gpus, 1) and rank == 0:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size wins\n")
    if torch.cuda.device_count() <= local_rank:          # counting devices does not initialise the GPU
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank} but this node shows {torch.cuda.device_count()} (there is no CPU fallback for the render path)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the render path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- scene (built on the host by this repo's own tools; identical on every rank) ----------------
    if args.workload == "atrium":
        sc = host.atrium_scene(args.width, args.height, bounces=args.bounces, detail=args.detail, accel=args.accel)
        workload = (f"atrium-{sc.blas[0].source_triangle_count}tri Sponza-class stand-in (sponza.obj absent from the reference mount), "
                    f"{args.accel.upper()} ({len(sc.blas[0].tri_hot)} refs, {len(sc.blas[0].nodes)} nodes), "
                    f"{args.width}x{args.height}, 3 lights, reflect/refract depth {args.bounces}, anisotropic mips")
    else:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import util
        name = {"monkey": "monkey", "tori16": "tori16", "materials": "materials_aniso"}[args.workload]
        sc, _ = util.load_golden(name)
        workload = f"{name} golden scene {sc.width}x{sc.height}"

    W, H = sc.width, sc.height
    T = sc.tile_count
    first, stride, count = rank, world, api.group_layout(W, H, world, rank)[2]     # tile t belongs to GPU t mod N
    if args.simulate_ranks > 1 and world == 1:
        first, stride, count = 0, args.simulate_ranks, api.group_layout(W, H, args.simulate_ranks, 0)[2]
    if args.force_gather and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

    # Frames in flight: like any real-time renderer the benchmark keeps consecutive frames in flight on separate contexts
    # (own queues, framebuffer and streams each), so one frame's kernel tails overlap the next frame's start.  Every frame
    # is rendered and gathered completely; --frames-in-flight 1 gives the one-frame-at-a-time latency figure.
    nfl = max(1, args.frames_in_flight)
    # measured on MI355X (ms per cfg3 frame, overlap / serial): 1 frame in flight 3.70 / 4.02, 2: 3.26 / 3.13, 3: 3.61 / 2.81, 4: 3.64 / 3.06
    serial = args.serial or (nfl > 1 and not args.overlap)

    class Slot:
        pass
    slots = []
    for k in range(nfl):
        sl = Slot()
        sl.r = api.Renderer(sc, device=local_rank)
        sl.fb_rgb = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        sl.fb_packed = torch.zeros(H * W, dtype=torch.int32, device="cuda")
        sl.r.bind_framebuffer(sl.fb_rgb.data_ptr(), sl.fb_packed.data_ptr())
        prio = os.environ.get("RTX_BENCH_STREAM_PRIORITIES", "")          # A/B runs: e.g. "-1,0,0" (torch: -1 = high priority)
        prio = int(prio.split(",")[k]) if prio and k < len(prio.split(",")) else 0
        sl.stream = torch.cuda.current_stream() if nfl == 1 else torch.cuda.Stream(priority=prio)
        sl.r.set_stream(sl.stream.cuda_stream)
        sl.gather = None
        if world > 1 or args.force_gather:
            # multi-GPU path of the C ABI (include/rtx.h rtx_group_*): RCCL is called by librtx_hip.so itself; torch.distributed only
            # carries the 128-byte communicator id from rank 0 to the other ranks (and the barrier / max-over-ranks of the timing contract)
            # ONE communicator per rank: the first frame slot creates it, the other frames in flight join it (rtx_group_attach)
            if k == 0:
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(api.Renderer.group_unique_id()), dtype=torch.uint8))
                if world > 1:
                    dist.broadcast(uid, src=0)
                sl.r.group_create(rank, world, bytes(uid.cpu().numpy().tobytes()))
            else:
                sl.r.group_attach(slots[0].r)
            sl.gather = True
        slots.append(sl)
    r, fb_rgb, fb_packed, gather = slots[0].r, slots[0].fb_rgb, slots[0].fb_packed, slots[0].gather
    frame_no = [0]

    def step():
        sl = slots[frame_no[0] % nfl]
        frame_no[0] += 1
        if sl.gather is not None:
            sl.r.group_render(cull_dead_shadow_rays=args.cull)            # this rank's tiles + ncclGather + frame assembly on rank 0, one stream
        else:
            sl.r.render_async(first, stride, count, simple_trace=args.simple_trace, cull_dead_shadow_rays=args.cull, serial=serial)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- untimed: ray counts and work counters of one frame -----------------------------------------------
    r.render_async(first, stride, count, count_work=True)
    stats, work = r.stats()
    rays_rank = sum(stats.values())
    tot = torch.tensor([stats["primary"], stats["shadow"], stats["reflection"], stats["refraction"], rays_rank], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(tot)
    tot = tot.cpu().tolist()
    level0_rays = stats["primary"]

    for _ in range(args.warmup):
        step()
    sync()
    # per-kernel HIP events (on the launch stream) over the timed region — on ONE of the contexts in flight: every frame that context renders
    # is timed kernel by kernel, the other contexts' frames carry no events (events around all 13 launches of every frame cost 3.5 % of the frame rate)
    timed_slots = slots[:1]
    first_timed_frame = frame_no[0]
    for sl in timed_slots:
        sl.r.enable_timing(not args.no_kernel_timing)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    t1 = time.perf_counter()
    ktimes = []
    for sl in timed_slots:
        ktimes += sl.r.kernel_times()
        sl.r.enable_timing(False)
    frames_with_events = sum(1 for f in range(first_timed_frame, first_timed_frame + args.steps) if f % nfl == 0)
    # extras (not the headline): (a) the same pipeline with dead shadow rays counted but not traversed; (b) one frame at a time
    t_cull = t_single = None
    if world == 1 and not args.cull and not args.simple_trace and gather is None and not args.no_extras:
        def timed(fn, n):
            for _ in range(3):
                fn()
            sync()
            a = time.perf_counter()
            for _ in range(n):
                fn()
            sync()
            return 1000.0 * (time.perf_counter() - a) / n

        def step_cull():
            sl = slots[frame_no[0] % nfl]; frame_no[0] += 1
            sl.r.render_async(first, stride, count, cull_dead_shadow_rays=True, serial=serial)
        t_cull = timed(step_cull, args.steps)
        t_single = timed(lambda: r.render_async(first, stride, count), min(args.steps, 15))    # slot 0 only, intra-frame overlap
        for sl in slots:                                # leave un-culled frames in the framebuffers for the parity check
            sl.r.render_async(first, stride, count)
        sync()
    # extra: per-kernel durations with every kernel on one stream (no overlap between shadow-ray and closest-hit kernels)
    serial_times = None
    if world == 1 and gather is None and not args.no_extras:
        r.enable_timing(True)
        for _ in range(min(args.steps, 10)):
            r.render_async(first, stride, count, serial=True)
        sync()
        serial_times = r.kernel_times()
        r.enable_timing(False)
    # the gathered frame on rank 0 must equal a frame rendered by rank 0 alone (untimed check)
    gather_ok = None
    if gather is not None and rank == 0:
        gathered = fb_packed.clone()                    # rank 0's packed framebuffer holds the assembled frame of the last group render
        r.render_async(0, 1, T)
        torch.cuda.synchronize()
        gather_ok = bool(torch.equal(gathered, fb_packed))
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ms_per_step = 1000.0 * elapsed / max(args.steps, 1)

    if rank == 0:
        mrays = tot[4] / (ms_per_step * 1e-3) * 1e-6
        steps = max(frames_with_events, 1)              # frames whose launches carried events
        kern = {k: {"launches_per_frame": len(v) / steps, "avg_ms": float(np.mean(v)), "total_ms_per_step": float(np.sum(v)) / steps} for k, v in _group(ktimes).items()}
        alone = None if not serial_times else {k: float(np.sum(v)) / min(args.steps, 10) for k, v in _group(serial_times).items()}      # ms per frame, this kernel alone on the chip
        roof = build_roofline(args, sc, work, stats, level0_rays, kern, alone, ms_per_step, tot)
        out = {
            "metric": "Mrays/s (primary+shadow+reflection+refraction)", "value": round(mrays, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "bounces": int(sc.config["bounces"][0]),
                       "triangles": sc.triangle_refs(), "bvh_nodes": int(sum(len(b.nodes) for b in sc.blas)),
                       "frames_in_flight": nfl, "streams_per_frame": 1 if serial else 2, "frames_timed_per_kernel": frames_with_events, "tiles": T, "sharding": "tile t -> GPU t mod N, tile-major buffers, one ncclGather per frame to rank 0 (rtx_group_render)" if world > 1 else "single GPU"},
            "rays_per_frame": {"primary": tot[0], "shadow": tot[1], "reflection": tot[2], "refraction": tot[3], "total": tot[4]},
            "mprimary_per_s": round(tot[0] / (ms_per_step * 1e-3) * 1e-6, 3),
            "kernels": kern,
            "gathered_frame_equals_single_gpu_frame": gather_ok,
            "kernels_serial_mode": None if not serial_times else {k: {"launches_per_frame": len(v) / min(args.steps, 10), "avg_ms": round(float(np.mean(v)), 5), "total_ms_per_frame": round(float(np.sum(v)) / min(args.steps, 10), 4)}
                                                                    for k, v in _group(serial_times).items()},
            "latency_ms_one_frame_at_a_time": None if t_single is None else round(t_single, 4),
            "extra_cull_dead_shadow_rays": None if t_cull is None else {"ms_per_step": round(t_cull, 4), "note": "RTX_RENDER_CULL_DEAD_SHADOW_RAYS: identical frame and ray counts; shadow rays with an exactly-zero light contribution are counted but not traversed"},
            "roofline": roof,
        }
        # ---- CPU baseline: the oracle (port of the reference's lane-1 path) on this box's host cores ------
        if world == 1 and args.cpu_seconds > 0:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import orc
            cores = baseline_threads()
            o = orc.OracleScene(sc)
            frames, t_cpu, ref = 0, 0.0, None
            while t_cpu < args.cpu_seconds and frames < 50:
                t = time.perf_counter()
                ref = o.render(threads=cores)
                t_cpu += time.perf_counter() - t
                frames += 1
            cpu_rays = sum(ref["stats"].values())
            port = {"value": round(cpu_rays * frames / t_cpu * 1e-6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                    "sample": f"{frames} full frame(s) of the same workload, oracle/rt_oracle.c (scalar lane-1 restatement), {cores} threads",
                    "ms_per_frame": round(1000.0 * t_cpu / frames, 2), "container_cpu_quota_cores": cpu_quota()}
            real = None if args.no_reference_baseline else reference_cpu_baseline(args, tot[4], stats)
            out["cpu_baseline"] = real or port                              # the real reference when its binary is here, else the port
            if real:
                out["cpu_baseline_port"] = port
            rgb = fb_rgb.cpu().numpy().reshape(H, W, 3)
            d = np.abs(rgb.astype(np.float64) - ref["rgb"].astype(np.float64))
            out["parity"] = {"max_abs_rgb_vs_oracle": float(np.nanmax(d)), "pixels_differing": int((d.max(axis=-1) > 0).sum()),
                             "packed_mismatch": int((fb_packed.cpu().numpy().view(np.uint32).reshape(H, W) != ref["packed"]).sum()),
                             "stats_equal": ref["stats"] == stats}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
