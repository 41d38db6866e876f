"""GPU: bench.py's output contract - ONE JSON line on stdout (libraries' chatter goes to stderr), the keys the driver reads, the
roofline object of the dominant kernel class, the workload named; a short run of the real script as a child process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-decode",
                          "--no-other-modes"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry the JSON line alone, got {len(lines)} lines"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "audio-frames/s"
    assert "configs[1]" in d["config"]["workload"] and d["config"]["parallelism"] == "dp1"
    assert d["value"] > 0 and abs(d["value"] - 32 * 998 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]     # frames of the batch per step time
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == pytest.approx(2500.0 / 6)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1
    assert r["launches_per_step"] > 100 and r["algorithmic_gflop_per_step"] > 1000
