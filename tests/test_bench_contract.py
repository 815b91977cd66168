"""CPU: the bench.py contract -- flags the driver passes, and the shape of the JSON line (checked on the line committed
under profiles/ by the last profiled run, which bench.py itself printed on an MI355X)."""

import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_the_driver_passes_exist():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\("%s", type=int' % flag, src), flag


def test_committed_json_line_has_the_contract_fields():
    lines = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_plain.json"))
    d = json.load(open(os.path.join(ROOT, "profiles", lines[-1])))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["data"] == "synthetic"
    # round 2 on: strong scaling of BASELINE.json's global env counts is the default (weak only with --envs)
    assert d["scaling"] in ("weak", "strong")
    if d["scaling"] == "strong":
        assert d["config"]["global_envs"] == d["config"]["envs_per_gpu"] * d["n_gpus"] == 65536
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert d["metric"].split()[0] == base["metric"].split()[0]
    assert d["n_gpus"] == 1 and d["steps"] > 0 and d["warmup"] >= 0
    assert abs(d["value"] - d["config"]["global_envs"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 5e-4 * r["frac"] and 0.0 < r["frac"] < 1.0   # (a line may carry 4 significant digits)
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic FLOPs per launch / average launch duration
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 1e-3 * r["achieved"]
    # round 5: the line stays under 4 KB and carries the strict-fp32 engine and the small shards inside `roofline`
    if "fp32_engine" in r:
        assert len(json.dumps(d)) < 4096 and r["fp32_engine"]["value"] > 0 and r["envs_4096"]["value"] > 0 and r["envs_8192"]["value"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_default_shards_are_the_baseline_configs():
    """--gpus N without --envs splits BASELINE.json's global env count (configs[4]: 65 536 G1-Walk over 8; configs[3]:
    32 768 humanoid over 4) with distributed.shard_bounds -> 8 192 envs per GPU."""
    import importlib.util
    import sys

    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from humanoid_amp_amd.distributed import shard_bounds

    assert bench.DEFAULT_GLOBAL_ENVS == {"g1_walk": 65536, "g1_dance": 65536, "humanoid3": 32768}
    for wl, world in (("g1_walk", 8), ("humanoid3", 4)):
        sizes = {shard_bounds(bench.DEFAULT_GLOBAL_ENVS[wl], world, r)[1] - shard_bounds(bench.DEFAULT_GLOBAL_ENVS[wl], world, r)[0]
                 for r in range(world)}
        assert sizes == {8192}
    assert bench.BASELINE_CONFIG[("g1_walk", 65536, 8)] == "configs[4]" and bench.BASELINE_CONFIG[("humanoid3", 32768, 4)] == "configs[3]"
