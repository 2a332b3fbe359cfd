"""Engine-level data parallelism: two ranks, each running the real HIP engine on its shard of the
(image, prompt) pairs behind parallel.DataParallelVQA, must return exactly what one process returns for the whole batch
(samples are independent rows of every kernel: qwen2_navit.py:602-614).  The ranks are spawned with the same launcher
`python bench.py --gpus N` uses."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_engine_worker.py")


def _last_json(text):
    for ln in reversed(text.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError(text[-2000:])


def test_two_engine_ranks_equal_one_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, WORKER], capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    single = _last_json(one.stdout)
    driver = tmp_path / "driver.py"
    driver.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom unimedvl_amd.launch import spawn_ranks\n"
                      f"sys.exit(spawn_ranks([sys.executable, {WORKER!r}], 2, timeout=500))\n")
    two = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=600, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    dp = _last_json(two.stdout)
    assert single["world"] == 1 and dp["world"] == 2
    assert single["calls"] == [5, 5] and dp["calls"] == [3, 3]          # rank 0 owns 3 of the 5 items either way
    assert len(single["out"]) == 5 and all(len(s.split()) == 6 for s in single["out"])
    assert dp["out"] == single["out"], "sharded run differs from the single-process run"
    assert dp["balanced"] == single["out"], "length-balanced assignment must return the same answers in the original order"
