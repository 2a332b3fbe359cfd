"""Data-parallel sharding / gather logic on CPU with the gloo backend, world_size 2
(the same code runs over RCCL on the GPUs; the path has no data-path collective)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unimedvl_amd.parallel import DataParallelVQA, all_gather_ragged, shard_bounds
    try:
        # 1) contiguous sharding of 5 (image, prompt) pairs over 2 ranks, gathered in order
        def engine(images, prompts):
            return [f"r{rank}:{im}:{p}" for im, p in zip(images, prompts)]
        dp = DataParallelVQA(engine)
        imgs, prompts = [f"img{i}" for i in range(5)], [f"q{i}" for i in range(5)]
        out = dp(imgs, prompts)
        expect = [f"r{0 if i < 3 else 1}:img{i}:q{i}" for i in range(5)]
        assert out == expect, out
        # 2) length-balanced assignment covers every item exactly once
        lens = [9, 1, 7, 3, 5, 2]
        out = dp([f"i{i}" for i in range(6)], [f"p{i}" for i in range(6)], lengths=lens)
        assert [o.split(":", 1)[1] for o in out] == [f"i{i}:p{i}" for i in range(6)]
        # 3) ragged all-gather of token ids ([steps_r, B_r] differs per rank), C1 of SURVEY.md
        local = torch.arange((rank + 2) * (3 - rank)).reshape(rank + 2, 3 - rank) + 100 * rank
        parts = all_gather_ragged(local)
        assert [tuple(p.shape) for p in parts] == [(2, 3), (3, 2)]
        assert torch.equal(parts[rank], local)
        assert torch.equal(parts[0], torch.arange(6).reshape(2, 3))
        # 4) more ranks than items
        s, e = shard_bounds(1, world, rank)
        assert (e - s) == (1 if rank == 0 else 0)
        assert dp(["a"], ["b"]) == ["r0:a:b"]
        ret[rank] = "ok"
    except Exception as ex:  # surface the failure in the parent
        ret[rank] = f"fail: {type(ex).__name__}: {ex}"
    finally:
        dist.destroy_process_group()


def test_dp_two_ranks_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "worker crashed or timed out"
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_shard_bounds_cover_everything():
    from unimedvl_amd.parallel import balanced_order, shard_bounds
    for n in (0, 1, 7, 8, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
    per = balanced_order([5, 1, 9, 3, 7, 2, 8], 3)
    assert sorted(i for p in per for i in p) == list(range(7))
