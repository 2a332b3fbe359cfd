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
