"""`python bench.py --gpus N` must bring up its N ranks itself (BASELINE configs[3]); checked here without a GPU through --dry-launch."""
import json
import os
import subprocess
import sys

from util import REPO


def _dry(*extra, env=None):
    e = dict(os.environ); e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-launch", *extra], env=e, capture_output=True, text=True, timeout=120, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def test_self_launch_plans_one_process_per_gpu():
    for n in (2, 8):
        j = _dry("--gpus", str(n), "--steps", "7", "--warmup", "2")
        assert j["n_gpus"] == n and len(j["ranks"]) == n
        ports = set()
        for r, p in enumerate(j["ranks"]):
            assert p["rank"] == r and p["device"] == r
            env = p["env"]
            assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == str(n)
            assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
            ports.add(env["MASTER_PORT"])
            assert p["argv"][1].endswith("bench.py") and "--dry-launch" not in p["argv"]
            assert p["argv"][2:] == ["--gpus", str(n), "--steps", "7", "--warmup", "2"]      # every rank runs the same benchmark arguments
        assert len(ports) == 1


def test_single_gpu_and_launcher_environments_start_nothing():
    assert _dry()["ranks"] == [] and _dry("--gpus", "1")["ranks"] == []
    # under torch.distributed.run bench.py IS a rank: it must not start ranks of its own
    j = _dry("--gpus", "4", env={"WORLD_SIZE": "4", "RANK": "1", "LOCAL_RANK": "1"})
    assert j["ranks"] == [] and j["n_gpus"] == 4


def test_master_port_is_taken_from_the_environment_when_given():
    j = _dry("--gpus", "2", env={"MASTER_PORT": "23456"})
    assert all(p["env"]["MASTER_PORT"] == "23456" for p in j["ranks"])


def test_the_parent_never_touches_the_gpu():
    """the self-launching parent must not import torch or load the HIP library: a process that has initialised the GPU may not be replaced or forked from on this pool"""
    src = open(os.path.join(REPO, "bench.py")).read()
    launcher = src[src.index("def launch_plan"):src.index("def main():")]
    assert "import torch" not in launcher and "pyrtx" not in launcher and "ctypes" not in launcher
    top_level = [l for l in src.splitlines() if l.startswith(("import ", "from "))]
    assert not any("torch" in l or "pyrtx" in l for l in top_level), top_level
