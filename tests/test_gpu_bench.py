"""bench.py end to end on the GPU box: the one-GPU line and the N>1 line (two ranks rehearsing on the box's single GPU,
exchanges staged through gloo) carry every field the bench contract and SURVEY 8(d-e) ask for."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print ONE line on stdout, got %d" % len(lines)
    return json.loads(lines[0]), p.stderr.decode()


def _check_roofline(r):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 * max(1.0, r["frac"])


def _check_cpu(c):
    assert c and c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"] and c["unit"]


def test_one_gpu_line_has_the_contract_fields():
    d, _ = _run(["--steps", "6", "--warmup", "2", "--no-north-star", "--no-configs", "--no-convergence"])
    assert d["metric"].startswith("training ratings/sec + val RMSE, MovieLens-1M SVD dim=64") and d["unit"] == "ratings/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 6, 2) and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    _check_roofline(d["roofline"])
    assert d["roofline"]["traffic"] is not None, "the committed PMC summary for this workload must be found"
    _check_cpu(d["cpu_baseline"])
    assert set(d["feeds"]) == {"device_drawn_ids", "host_drawn_ids", "prestaged_ids"}


def test_two_rank_line_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts the ranks; the line is the data-parallel headline with
    cpu_baseline, the single-GPU reference and BASELINE config 4 row-sharded over the two ranks (tables scaled down for the
    shared GPU)."""
    d, err = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--c4-scale", "100"],
                  {"TFR_DIST_BACKEND": "gloo", "TFR_SHARE_GPU": "1"})
    assert "started 2 ranks" in err and "world size 2" in err
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["unit"] == "ratings/s"
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    _check_roofline(d["roofline"])
    assert d["roofline"]["bound"] == "xgmi" and "phases_us" in d["roofline"]
    _check_cpu(d["cpu_baseline"])
    assert d["single_gpu_reference"]["value"] > 0
    s = d["sharded_c4"]
    assert s["value"] > 0 and s["ms_per_step"] > 0 and s["config"]["global_batch"] == 2 * 262144
    _check_roofline(s["roofline"])
    assert s["roofline"]["bound"] == "xgmi" and s["phases_us"] and "gather" in s["phases_us"]
    assert s["single_gpu_reference"]["value"] > 0
    _check_cpu(s["cpu_baseline"])
