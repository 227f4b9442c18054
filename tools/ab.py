#!/usr/bin/env python3
"""A/B of librtx_hip tuning variants: runs bench.py once per variant (separate processes), prints ms/frame and kernel totals."""
import json, os, subprocess, sys
variants = sys.argv[1:]
for v in variants:
    env = dict(os.environ)
    extra = []
    if ":" in v:
        v, flag = v.split(":", 1); extra = ["--" + flag]
    if v != "base":
        env["RTX_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd", "csrc", f"librtx_hip_{v}.so")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-seconds", "0"] + [e for e in extra if e != "--extras"] + ([] if "--extras" in extra else ["--no-extras"]), capture_output=True, text=True, env=env)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(v, extra, j["ms_per_step"], j["value"], {k: round(x["total_ms_per_step"], 3) for k, x in j["kernels"].items()},
              "serial:", {k: x["total_ms_per_frame"] for k, x in (j.get("kernels_serial_mode") or {}).items()}, "latency1:", j.get("latency_ms_one_frame_at_a_time"), flush=True)
    except Exception as e:
        print(v, "FAILED", out.stderr[-400:], flush=True)
