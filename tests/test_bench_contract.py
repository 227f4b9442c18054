"""The bench line's contract (driver side): required keys, types and the two extra objects, checked on the committed artefact of the
latest run (profiles/r01_u_bench_atrium.json, produced by `python bench.py` on an MI355X) and on bench.py's own argument defaults."""
import json
import os
import re

from util import REPO


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(REPO, "profiles", "r01_u_bench_atrium.json")).read().strip().splitlines()[-1]
    j = json.loads(line)
    for k, t in {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
                 "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict, "cpu_baseline": dict}.items():
        assert isinstance(j[k], t), k
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["scaling"] == "strong" and j["higher_is_better"] is True
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
    assert j["parity"] == {"max_abs_rgb_vs_oracle": 0.0, "pixels_differing": 0, "packed_mismatch": 0, "stats_equal": True}
    assert abs(j["value"] - j["rays_per_frame"]["total"] / (j["ms_per_step"] * 1e-3) * 1e-6) / j["value"] < 1e-3


def test_bench_defaults_are_single_gpu_and_short():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src) and re.search(r'"--steps", type=int, default=30', src) and re.search(r'"--warmup", type=int, default=5', src)
    body = src.split("# ---- CPU baseline")[0]
    assert "import orc" not in body                                      # nothing before the baseline leg imports the checker
