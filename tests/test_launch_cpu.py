"""unimedvl_amd.launch.spawn_ranks (what `python bench.py --gpus N` uses when it is not under torchrun): every rank gets
the torch.distributed environment, rank 0 owns stdout, a failing rank takes the job down instead of hanging it."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, world, timeout=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from unimedvl_amd.launch import spawn_ranks
        sys.exit(spawn_ranks([sys.executable, {str(script)!r}], {world}, timeout={timeout!r}))
    """))
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=300)


def test_spawn_ranks_gloo_allgather_one_stdout_line(tmp_path):
    r = _run(tmp_path, """
        import os, torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
        assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
        dist.init_process_group("gloo")
        parts = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, torch.tensor([rank, world]))
        dist.barrier()
        print("line from rank", rank, [p.tolist() for p in parts], flush=True)
        dist.destroy_process_group()
    """, 2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]     # gloo's own connect banner
    assert lines == ["line from rank 0 [[0, 2], [1, 2]]"], (r.stdout, r.stderr[-500:])     # rank 0 owns stdout
    assert "line from rank 1 [[0, 2], [1, 2]]" in r.stderr                                  # the others go to stderr


def test_spawn_ranks_failure_is_propagated_not_hung(tmp_path):
    r = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(60)      # would hang in a collective forever: the launcher must terminate it
    """, 2)
    assert r.returncode == 3


def test_bench_argument_contract():
    """bench.py keeps the driver's CLI (--gpus / --steps / --warmup) and refuses to run without a GPU instead of falling back"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)


def test_cpu_binding_plan_follows_the_gpu_socket(tmp_path):
    """one process per GPU on a 2-socket host (8 GPUs, 4 per socket): every rank gets a disjoint share of ITS socket's CPUs"""
    from unimedvl_amd.launch import gpu_numa_node, parse_cpulist, plan_cpu_binding, share_of
    assert parse_cpulist("0-3,8-11\n") == [0, 1, 2, 3, 8, 9, 10, 11] and parse_cpulist("5") == [5] and parse_cpulist("") == []
    assert share_of(list(range(10)), 0, 3) == [0, 1, 2] and share_of(list(range(10)), 2, 3) == [6, 7, 8, 9]
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = list(range(256))
    plans = [plan_cpu_binding(r, 8, numa, allowed, node_cpus) for r in range(8)]
    for r, cpus in enumerate(plans):
        assert len(cpus) == 32 and set(cpus) <= set(node_cpus[numa[r]])
    assert len(set().union(*map(set, plans))) == 256                     # disjoint and complete
    # an affinity mask the job already has (cgroup, taskset) is respected
    assert set(plan_cpu_binding(1, 8, numa, list(range(0, 32)), node_cpus)) <= set(range(32))
    # no NUMA information: an even contiguous split
    assert plan_cpu_binding(3, 4, [-1, -1, -1, -1], list(range(16)), {}) == [12, 13, 14, 15]
    # sysfs lookup
    d = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    d.mkdir(parents=True)
    (d / "numa_node").write_text("1\n")
    assert gpu_numa_node("0000:C1:00.0", sysfs=str(tmp_path)) == 1 and gpu_numa_node("0000:aa:00.0", sysfs=str(tmp_path)) == -1
