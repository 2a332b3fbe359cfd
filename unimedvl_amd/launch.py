"""One process per GPU on one node, without torchrun: what `python bench.py --gpus N` does when it is not already running
under `torch.distributed.run` (data parallel, one rank per GPU, rendezvous on 127.0.0.1).

The reference has no launcher (it is single-process); this is the smallest thing that gives every
rank the environment `torch.distributed.init_process_group` reads (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).
"""
import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    return env


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8-11' (the format of /sys/devices/system/node/nodeN/cpulist) -> [0, 1, 2, 3, 8, 9, 10, 11]"""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def share_of(cpus: Sequence[int], index: int, count: int) -> List[int]:
    """The index-th of `count` contiguous, near-equal shares of a CPU list (never empty when there are >= count CPUs)."""
    cpus = list(cpus)
    if count <= 1 or len(cpus) < count:
        return cpus
    lo, hi = index * len(cpus) // count, (index + 1) * len(cpus) // count
    return cpus[lo:hi]


def gpu_numa_node(pci_bdf: str, sysfs: str = "/sys") -> int:
    """NUMA node of a PCI device ('0000:c1:00.0'); -1 when the platform does not say."""
    try:
        return int(open(os.path.join(sysfs, "bus", "pci", "devices", pci_bdf.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1


def plan_cpu_binding(local_rank: int, local_world: int, numa_of_rank: Sequence[int], allowed: Sequence[int],
                     node_cpus: Optional[dict] = None) -> List[int]:
    """CPUs for one rank of a one-process-per-GPU job on a multi-socket host (MI355X nodes: 2 x 64 cores, 4 GPUs per socket):
    the CPUs of the GPU's own NUMA node, divided among the ranks whose GPUs sit on that node, so that the host side of a rank
    (tokeniser, image transforms, launch loop, the pinned staging buffers it allocates) stays on the socket its GPU hangs off.
    Without NUMA information: an even contiguous split of the allowed CPUs."""
    allowed = sorted(allowed)
    node = numa_of_rank[local_rank] if local_rank < len(numa_of_rank) else -1
    if node is not None and node >= 0 and node_cpus and node in node_cpus:
        mine = [c for c in node_cpus[node] if c in set(allowed)]
        peers = [r for r in range(local_world) if r < len(numa_of_rank) and numa_of_rank[r] == node]
        if mine and local_rank in peers:
            return share_of(mine, peers.index(local_rank), len(peers))
    return share_of(allowed, local_rank, local_world)


def bind_rank_to_gpu_socket(local_rank: int, local_world: int, device_index: Optional[int] = None) -> dict:
    """Pin the calling process to the CPUs next to its GPU (Linux; a no-op description elsewhere).  Returns what was done -
    bench.py reports it in config.cpu_binding.  Respects an affinity mask the process already has (cgroups, taskset)."""
    info = {"bound": False}
    if local_world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("UMV_NO_CPU_BINDING"):
        return info
    try:
        import torch
        allowed = sorted(os.sched_getaffinity(0))
        numa = []
        for r in range(local_world):
            try:
                idx = device_index if (device_index is not None and r == local_rank) else r % max(1, torch.cuda.device_count())
                p = torch.cuda.get_device_properties(idx)
                numa.append(gpu_numa_node("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)))
            except Exception:
                numa.append(-1)
        node_cpus = {}
        base = "/sys/devices/system/node"
        if os.path.isdir(base):
            for d in os.listdir(base):
                if d.startswith("node") and d[4:].isdigit():
                    try:
                        node_cpus[int(d[4:])] = parse_cpulist(open(os.path.join(base, d, "cpulist")).read())
                    except OSError:
                        pass
        cpus = plan_cpu_binding(local_rank, local_world, numa, allowed, node_cpus)
        if cpus:
            os.sched_setaffinity(0, cpus)
            torch.set_num_threads(max(1, min(len(cpus), 32)))
            info = {"bound": True, "numa_node": numa[local_rank] if local_rank < len(numa) else -1, "cpus": len(cpus),
                    "first_cpu": cpus[0], "last_cpu": cpus[-1]}
    except Exception as e:      # binding is an optimisation: never take the job down
        info = {"bound": False, "error": f"{type(e).__name__}: {e}"}
    return info


def spawn_ranks(argv: Sequence[str], world: int, port: Optional[int] = None, timeout: Optional[float] = None,
                poll: float = 0.2) -> int:
    """Run `argv` as `world` processes (rank r gets RANK = LOCAL_RANK = r).  Rank 0 inherits stdout, so whatever single
    line it prints is the job's output; the other ranks' stdout is folded into stderr.  Returns the largest exit code;
    if one rank fails the rest are terminated (exact PIDs) instead of hanging in a collective."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port() if port is None else port
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port),
                                      stdout=None if r == 0 else sys.stderr, stderr=None))
    t0 = time.time()
    rc = 0
    live = set(range(world))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = max(rc, code if code > 0 else 1)
                for o in live:
                    procs[o].terminate()
        if timeout is not None and time.time() - t0 > timeout:
            for o in live:
                procs[o].kill()
            rc = max(rc, 124)
            break
        if live:
            time.sleep(poll)
    for p in procs:
        try:
            p.wait(10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc
