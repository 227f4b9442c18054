"""The bench line's contract (driver side): required keys, types and the two extra objects, checked on the committed artefact of the
latest run (the newest profiles/r*_bench_atrium.json, produced by `python bench.py` on an MI355X through tools/profile_round.sh) and on
bench.py's own argument defaults."""
import json
import os
import re

from util import REPO


def test_committed_bench_line_has_the_contract_fields():
    import glob
    newest = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9][a-z]_bench_atrium.json")))[-1]
    line = open(newest).read().strip().splitlines()[-1]
    j = json.loads(line)
    for k, t in {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
                 "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict, "cpu_baseline": dict}.items():
        assert isinstance(j[k], t), k
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["scaling"] == "strong" and j["higher_is_better"] is True
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    # the bound is what binds (VALU issue): units of the algorithm per second against a ceiling measured in the same run
    assert r["bound"] == "valu" and r["kernel"] in ("k_trace_any", "k_trace_closest") and "equivalents/s" in r["unit"]
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0 and r["frac_one_frame_at_a_time"] <= 1.0
    assert r["hbm"]["peak_GBps"] == 8000.0 and 0.0 < r["hbm"]["traffic_frac_of_peak"] < 1.0
    assert os.path.exists(os.path.join(REPO, r["hbm"]["traffic_source"]))
    g = r["other"]["k_trace_closest"]["gather"]
    assert 0.0 < g["frac"] <= 1.0 and os.path.exists(os.path.join(REPO, g["ceiling_source"]))
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
    assert j["parity"] == {"max_abs_rgb_vs_oracle": 0.0, "pixels_differing": 0, "packed_mismatch": 0, "stats_equal": True}
    assert abs(j["value"] - j["rays_per_frame"]["total"] / (j["ms_per_step"] * 1e-3) * 1e-6) / j["value"] < 1e-3


def test_bench_defaults_are_single_gpu_and_short():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src) and re.search(r'"--steps", type=int, default=30', src) and re.search(r'"--warmup", type=int, default=5', src)
    body = src.split("# ---- CPU baseline")[0]
    assert "import orc" not in body                                      # nothing before the baseline leg imports the checker
