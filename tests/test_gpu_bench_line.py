"""GPU: the contract of bench.py's line (what the driver parses): ONE JSON object on the last stdout line, under 4 KB, with the
metric / config / roofline / cpu_baseline objects the measurement section of DESIGN.md names -- the strict-fp32 engine and the
4 096- / 8 192-env shards INSIDE `roofline`, the update's milliseconds inside `config` -- and the detail on stderr.  A short
run (3 steps, 2 048-env secondary runs skipped) started as a child process."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_is_one_small_json_object_with_the_key_numbers_inside_roofline_and_config():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--envs", "32768",
                        "--no-dropin", "--no-configs", "--cpu-envs", "512"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.strip()]
    line = lines[-1]
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["scaling"] == "weak"
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.0 < r["frac"] < 1.0 and r["kernel"] == "disc_mlp_fused_kernel"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    for inner in ("fp32_engine", "envs_4096", "envs_8192", "env_launch"):
        assert inner in r and (inner == "env_launch" or r[inner]["value"] > 0), inner
    assert r["fp32_engine"]["dtype"] == "f32" and 0.0 < r["fp32_engine"]["frac"] < 1.0
    assert r["env_launch"]["bound"] == "hbm" and 0.0 < r["env_launch"]["frac"] < 1.0
    c = d["config"]
    assert c["envs_per_gpu"] == 32768 and c["disc_plan"]["fused_rows"] == 32768 and c["collective"].startswith("none")
    u = c["update"]
    assert u["train_steps"] == 12 and u["global_minibatch"] == 4096 and u["ms_per_update"] > 0 and not u["in_timed_region"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    detail = [ln for ln in p.stderr.splitlines() if ln.startswith("[bench detail] ")]
    assert detail and "kernel_us_per_step" in json.loads(detail[-1][len("[bench detail] "):])
