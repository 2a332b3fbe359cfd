"""bench.py's one-line JSON contract (task statement, "Measurement"), exercised end to end on the tiny config so that a
change to the engine cannot silently break the line the driver parses; and __graft_entry__.smoke()."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*flags):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "6", "--warmup", "2", *flags],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "tokens/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 8 * 6 / (d["ms_per_step"] * 6e-3)) <= 0.02 * d["value"]        # value == tokens / timed seconds
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "t2i" in d and d["t2i"]["images_per_s"] > 0


def test_bench_no_graph_and_flags():
    d = _run_bench("--no-graph", "--no-t2i", "--batch", "3")
    assert d["config"]["decode"] == "eager" and d["config"]["batch_per_gpu"] == 3 and "t2i" not in d
    assert d["value"] > 0


def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
