"""Host logic of kvcache.PagedCache (the block-table KV cache) on CPU tensors: page accounting, the device table's
contents, release / re-use, reference-counted snapshots with copy-on-write of the shared last page, segment views, the pool bounds.
(The kernels that read the table are GPU tests: tests/test_paged_kv_gpu.py.)"""
import pytest
import torch

from unimedvl_amd import ops
from unimedvl_amd.kvcache import PagedCache

NKV, HD = 1, 8
P = ops.KV_PAGE


def _fill(c, seg, lo, hi, val):
    """write `val + position` into K / V^T of positions lo..hi-1 of a segment, through the table (what umv_qkv_post does on the device)"""
    for l, sl in enumerate(c.slabs):
        for pos in range(lo, hi):
            pg = int(sl.table[seg, pos // P])
            sl.k[pg, :, pos % P, :] = val + pos + 1000 * l
            sl.vt[pg, :, :, pos % P] = val + pos + 1000 * l


def test_pages_follow_the_context_and_come_back():
    c = PagedCache(2, pool_pages=12, max_context=2048)
    c.ensure_tokens([300, 10, 0], NKV, HD, "cpu")
    assert c.pages_in_use() == 2 + 1 + 0 and c.cap == 2048
    t = c.slabs[0].table
    assert t.shape == (3, 8) and len({int(t[0, 0]), int(t[0, 1]), int(t[1, 0])}) == 3 and 0 not in (int(t[0, 0]), int(t[0, 1]), int(t[1, 0]))
    c.lens = [300, 10, 0]
    c.ensure_tokens([300, 10, 0], NKV, HD, "cpu")               # nothing grows: nothing is taken
    assert c.pages_in_use() == 3
    c.ensure_tokens([513, 10, 256], NKV, HD, "cpu")             # 3 pages, 1 page, exactly one page
    assert c.pages_in_use() == 3 + 1 + 1
    with pytest.raises(ValueError):
        c.ensure_tokens([4000, 10, 256], NKV, HD, "cpu")        # beyond the table's reach
    with pytest.raises(RuntimeError):
        c.ensure_tokens([2048, 2048, 256], NKV, HD, "cpu")      # beyond the pool (11 usable pages)
    assert c.pages_in_use() == 5, "a refused request must not keep pages"
    used = c.pages_in_use()
    c.release(0)
    assert c.lens[0] == 0 and c.pages_in_use() < used
    c.release(1), c.release(2)
    assert c.pages_in_use() == 0
    c.ensure_tokens([8 * P, 3 * P, 0], NKV, HD, "cpu")          # every page of the pool can be taken again
    assert c.pages_in_use() == 11


def test_snapshot_shares_pages_and_copies_the_last_one_on_append():
    c = PagedCache(2, pool_pages=16, max_context=2048)
    c.ensure_tokens([300, 256], NKV, HD, "cpu")
    _fill(c, 0, 0, 300, 0.0), _fill(c, 1, 0, 256, 5000.0)
    c.lens = [300, 256]
    s = c.snapshot()
    assert c.pages_in_use() == 3 and s.lens == [300, 256]
    k0 = [c.packed_keys(l).clone() for l in range(2)]
    # the original appends: segment 0's half-filled last page is shared -> copied; segment 1 ends on a page boundary -> a fresh page
    c.ensure_tokens([310, 260], NKV, HD, "cpu")
    assert c.pages_in_use() == 3 + 1 + 1
    _fill(c, 0, 300, 310, 0.0), _fill(c, 1, 256, 260, 5000.0)
    c.lens = [310, 260]
    for l in range(2):
        assert torch.equal(s.packed_keys(l), k0[l]), "the snapshot must not see the original's appends"
        assert torch.equal(c.packed_keys(l)[:300], k0[l][:300]) and torch.equal(c.packed_values(l)[310:566], s.packed_values(l)[300:556])
    # the snapshot appends too: its last page of segment 0 is its own by now (the original left it), no further copy
    s.ensure_tokens([305, 256], NKV, HD, "cpu")
    assert c.pages_in_use() == 5
    _fill(s, 0, 300, 305, 9000.0)
    s.lens = [305, 256]
    assert torch.equal(c.packed_keys(0)[300:310, 0, 0], torch.arange(300, 310).float().to(c.packed_keys(0).dtype))
    # releasing both sides returns every page exactly once
    for seg in range(2):
        c.release(seg), s.release(seg)
    assert c.pages_in_use() == 0


def test_views_share_the_pool():
    c = PagedCache(1, pool_pages=10, max_context=1024)
    c.ensure_tokens([0, 0, 0], NKV, HD, "cpu")
    v = c.view_segments(1, 2)
    v.ensure_tokens([600], NKV, HD, "cpu")
    v.lens = [600]
    assert c.pages_in_use() == 3 and len(c.pool.host_table[1]) == 3 and c.pool.host_table[0] == []
    assert v.slabs[0].table.data_ptr() == c.slabs[0].table[1:2].data_ptr() and v.slabs[0].table.stride(0) == c.slabs[0].table.stride(0)
    c.lens[1] = v.lens[0]
    assert c.packed_keys(0).shape[0] == 600


def test_pool_larger_than_32_bit_offsets_is_refused():
    with pytest.raises(ValueError):
        c = PagedCache(1, pool_pages=9000, max_context=1024)
        c.ensure_tokens([1], 4, 128, "meta")
