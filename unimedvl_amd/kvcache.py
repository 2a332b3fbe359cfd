"""KV cache as in-place slabs.

Drop-in for the reference's ``NaiveCache`` (codes/modeling/unimedvl/qwen2_navit.py:207-221):
same constructor ``NaiveCache(num_layers)``, ``num_layers`` / ``seq_lens`` properties,
and it survives ``copy.deepcopy`` (the inferencer snapshots contexts that way,
codes/inferencer.py:587,600,607,261).  The reference re-allocates and re-scatters the
whole ``[sum K, kvh, hd]`` tensor on every forward (qwen2_navit.py:585-600); here each
layer owns  K [seg][kvh][cap][hd]  and  V^T [seg][kvh][hd][cap]  bf16 slabs that are
appended in place, with the per-sample lengths kept on the host.
"""
import torch

from . import ops


def _round_up(x, m):
    return (x + m - 1) // m * m


class NaiveCache:
    def __init__(self, num_layers):
        self._num_layers = num_layers
        self.slabs = None          # list[KVSlab] once materialised
        self.lens = []             # committed tokens per segment (host ints)
        self.nkv = self.hd = None
        self.device = None
        self._borrowed = False     # slabs belong to another cache (snapshot()): copy before the first write
        self.reserved = False      # reserve() was called: slab addresses are stable for the life of the cache

    # --- prefix sharing ("prefix-sharing of the three CFG contexts instead of deepcopy")
    def snapshot(self):
        """O(1) logical copy: shares this cache's slabs and freezes the current lengths.  Safe because tokens are only ever
        APPENDED in place beyond `lens` (this cache may go on appending or decoding; the snapshot never sees those slots)
        and because a snapshot copies itself before ITS first write (copy-on-write in ensure()).  Replaces the whole-KV
        ``deepcopy(gen_context)`` the reference takes per text item (inferencer.py:261,587,600,607)."""
        c = NaiveCache(self._num_layers)
        c.lens = list(self.lens)
        c.nkv, c.hd, c.device = self.nkv, self.hd, self.device
        c.slabs = self.slabs
        c._borrowed = self.slabs is not None
        return c

    def shares_storage_with(self, other):
        """True when both caches read the same slab memory (one is a snapshot of the other, or both of a third)."""
        if self.slabs is None or other.slabs is None or len(self.slabs) != len(other.slabs):
            return False
        return all(a.k.data_ptr() == b.k.data_ptr() and a.vt.data_ptr() == b.vt.data_ptr() and a.cap == b.cap
                   for a, b in zip(self.slabs, other.slabs))

    def _materialize(self, need_cap):
        """private copy of the visible prefix (what deepcopy would have made when the snapshot was taken)"""
        nseg = len(self.lens)
        cap = _round_up(max(need_cap, self.slabs[0].cap if self.slabs else 0, 256), 256)
        keep = _round_up(max(max(self.lens), 1), 32)
        new = []
        for old in self.slabs:
            s = ops.KVSlab(nseg, self.nkv, cap, self.hd, self.device)
            s.k[:, :, :keep].copy_(old.k[:, :, :keep])
            s.vt[:, :, :, :keep].copy_(old.vt[:, :, :, :keep])
            new.append(s)
        self.slabs = new
        self._borrowed = False

    # --- reference-compatible surface
    @property
    def num_layers(self):
        return self._num_layers

    @property
    def seq_lens(self):
        return sum(self.lens)

    @property
    def key_cache(self):
        """{layer: [sum K, kvh, hd]} view materialised on demand (tests / debugging)."""
        return {l: self.packed_keys(l) for l in range(self._num_layers)}

    @property
    def value_cache(self):
        return {l: self.packed_values(l) for l in range(self._num_layers)}

    def packed_keys(self, layer):
        if self.slabs is None or not any(self.lens):
            return None
        return torch.cat([self.slabs[layer].k[s, :, :n].transpose(0, 1) for s, n in enumerate(self.lens)], 0)

    def packed_values(self, layer):
        if self.slabs is None or not any(self.lens):
            return None
        return torch.cat([self.slabs[layer].vt[s, :, :, :n].permute(2, 0, 1) for s, n in enumerate(self.lens)], 0)

    # --- slab management
    @property
    def cap(self):
        return 0 if self.slabs is None else self.slabs[0].cap

    def ensure(self, nseg, need_cap, nkv, hd, device):
        """Make room for `need_cap` tokens per segment (committed + the call's new tokens)."""
        need_cap = _round_up(max(need_cap, 32), 32)
        if self.slabs is None:
            cap = _round_up(max(need_cap, 256), 256)
            self.slabs = [ops.KVSlab(nseg, nkv, cap, hd, device) for _ in range(self._num_layers)]
            self.lens = [0] * nseg
            self.nkv, self.hd, self.device = nkv, hd, device
            return
        if nseg != len(self.lens):
            raise ValueError(f"cache holds {len(self.lens)} samples, call has {nseg}")
        if self._borrowed:          # every forward / decode calls ensure() before it writes: copy-on-write happens here
            self._materialize(need_cap)
        if need_cap > self.cap:
            cap = _round_up(max(need_cap, 2 * self.cap), 256)
            keep = max(self.lens)
            new = []
            for old in self.slabs:
                s = ops.KVSlab(nseg, self.nkv, cap, self.hd, self.device)
                if keep:
                    s.k[:, :, :keep].copy_(old.k[:, :, :keep])
                    s.vt[:, :, :, :keep].copy_(old.vt[:, :, :, :keep])
                new.append(s)
            self.slabs = new

    def reserve(self, nseg, cap, nkv, hd, device):
        """Pre-size (bench / serving) so that decode never re-allocates.  A reserved cache keeps its slab addresses, which is
        what lets fixed-shape prefills into it replay from a HIP graph (bagel.Bagel.forward_cache_update_vit)."""
        self.ensure(nseg, cap, nkv, hd, device)
        self.reserved = True

    @staticmethod
    def merged(caches, nsegs, extra, nkv, hd, device):
        """One cache whose segments are the segments of `caches` back to back (contexts that hold
        different tokens for the same samples, e.g. the three CFG contexts of bagel.py:1120-1171),
        with room for `extra` more tokens per segment.  Lets independent passes run as ONE packed
        forward.  `nsegs[i]` = samples in caches[i] (needed for caches that are still empty)."""
        lens = []
        for c, n in zip(caches, nsegs):
            lens += list(c.lens) if c.slabs is not None else [0] * n
        m = NaiveCache(caches[0].num_layers)
        m.ensure(len(lens), max(lens) + extra, nkv, hd, device)
        m.lens = lens
        base = 0
        for c, n in zip(caches, nsegs):
            if c.slabs is not None and max(c.lens) > 0:
                keep = _round_up(max(c.lens), 32)
                for dst, src in zip(m.slabs, c.slabs):
                    dst.k[base:base + n, :, :keep].copy_(src.k[:, :, :keep])
                    dst.vt[base:base + n, :, :, :keep].copy_(src.vt[:, :, :, :keep])
            base += n
        return m

    def view_segments(self, start, end):
        """A cache over segments [start, end) sharing this cache's memory (no copy)."""
        if self._borrowed:
            self._materialize(self.cap)
        v = NaiveCache(self._num_layers)
        v.lens = list(self.lens[start:end])
        v.nkv, v.hd, v.device = self.nkv, self.hd, self.device
        v.slabs = [ops.KVSlab.from_tensors(s.k[start:end], s.vt[start:end]) for s in self.slabs]
        return v

    def __deepcopy__(self, memo):
        c = NaiveCache(self._num_layers)
        c.lens = list(self.lens)
        c.nkv, c.hd, c.device = self.nkv, self.hd, self.device
        if self.slabs is not None:
            keep = _round_up(max(max(self.lens), 1), 32)
            cap = self.cap
            c.slabs = []
            for old in self.slabs:
                s = ops.KVSlab(len(self.lens), self.nkv, cap, self.hd, self.device)
                s.k[:, :, :keep].copy_(old.k[:, :, :keep])
                s.vt[:, :, :, :keep].copy_(old.vt[:, :, :, :keep])
                c.slabs.append(s)
        return c


class _PagePool:
    """Allocator state shared by a PagedCache and its views / snapshots: the per-layer pools, the free list, page reference counts."""

    def __init__(self, num_layers, npages, nkv, hd, device, nseg, max_pages):
        if npages * nkv * ops.KV_PAGE * hd * 2 > 2 ** 31 - 1:       # the LDS-shared prefill kernels reach a page through a 32-bit byte offset
            raise ValueError(f"paged KV pool of {npages} pages x {nkv * ops.KV_PAGE * hd * 2} bytes exceeds 2 GiB per layer and operand")
        self.table = torch.zeros((nseg, max_pages), dtype=torch.int32, device=device)      # entry p of segment s: pool page of keys p*256..
        self.host_table = [[] for _ in range(nseg)]                                       # pages each segment owns, in order
        self.slabs = [ops.PagedSlab(npages, nkv, hd, device, self.table) for _ in range(num_layers)]
        self.free = list(range(npages - 1, 0, -1))       # page 0 stays unused: a table entry of 0 is "no page" (reads of it are masked anyway)
        self.refs = [0] * npages
        self.npages, self.nkv, self.hd, self.device = npages, nkv, hd, device

    def alloc(self):
        if not self.free:
            raise RuntimeError(f"paged KV pool exhausted ({self.npages - 1} pages of {ops.KV_PAGE} tokens): raise pool_pages")
        p = self.free.pop()
        self.refs[p] = 1
        return p

    def unref(self, p):
        self.refs[p] -= 1
        if self.refs[p] == 0:
            self.free.append(p)


class PagedCache:
    """Block-table KV cache: every layer owns ONE pool of 256-token pages, K [page][kvh][256][hd] and
    V^T [page][kvh][hd][256], that all segments draw from; a segment's context is the list of its pages (`table` on the device).
    A context grows by taking pages - nothing is copied or re-allocated, the pool is shared by short and long requests - and a finished
    request returns its pages (`release`).  snapshot() shares the pages of the prefix (reference counts; the partially filled last page is
    copied when the snapshot or the original appends to it).

    Same surface as NaiveCache for the paths that take it: LanguageModel.forward_inference (prefill: umv_qkv_post writes through the
    table, the attention of a paged call runs on the per-wave kernel), decode.DecodeSession (the captured step reads the table from
    device memory: pages for the whole decode horizon are taken before the capture) and serving.ContinuousBatcher(paged=True).  The
    reference's NaiveCache is unbounded because it re-merges the cache on every forward (qwen2_navit.py:585-600)."""

    def __init__(self, num_layers, pool_pages=1024, max_context=32768):
        self._num_layers = num_layers
        self.pool_pages, self.max_pages = int(pool_pages), (int(max_context) + ops.KV_PAGE - 1) // ops.KV_PAGE
        self.pool = None
        self.lens = []
        self._seg0 = 0                      # view_segments(): first segment of the pool's table this cache covers
        self.nkv = self.hd = self.device = None
        self.reserved = False               # (no HIP-graph prefills into a paged cache: the table changes between calls)

    # --- NaiveCache-compatible surface
    @property
    def num_layers(self):
        return self._num_layers

    @property
    def seq_lens(self):
        return sum(self.lens)

    @property
    def cap(self):
        return self.max_pages * ops.KV_PAGE

    @property
    def slabs(self):
        if self.pool is None:
            return None
        if self._seg0 == 0 and len(self.lens) == self.pool.table.shape[0]:
            return self.pool.slabs
        # a view: the same pools behind a slice of the table rows
        t = self.pool.table[self._seg0:self._seg0 + len(self.lens)]
        out = []
        for s in self.pool.slabs:
            v = ops.PagedSlab.__new__(ops.PagedSlab)
            v.k, v.vt, v.table, v.nkv, v.hd, v.cap = s.k, s.vt, t, s.nkv, s.hd, s.cap
            out.append(v)
        return out

    def _pages(self, seg):
        return self.pool.host_table[self._seg0 + seg]

    def ensure(self, nseg, need_cap, nkv, hd, device):
        """NaiveCache.ensure's contract: room for need_cap tokens in EVERY segment (callers that know per-segment needs use ensure_tokens)"""
        self.ensure_tokens([need_cap] * nseg, nkv, hd, device)

    def ensure_tokens(self, need, nkv, hd, device):
        """Pages for need[s] tokens in segment s (committed + the call's new tokens); only segments that grow take pages."""
        nseg = len(need)
        if self.pool is None:
            self.pool = _PagePool(self._num_layers, self.pool_pages, nkv, hd, device, nseg, self.max_pages)
            self.lens = [0] * nseg
            self.nkv, self.hd, self.device = nkv, hd, device
        if nseg != len(self.lens):
            raise ValueError(f"cache holds {len(self.lens)} samples, call has {nseg}")
        changed = []
        want = 0                              # all-or-nothing: check the reach and the pool before taking a single page
        for s, n in enumerate(need):
            if n > self.cap:
                raise ValueError(f"segment {s}: {n} tokens exceed the page table's reach of {self.cap} (max_context)")
            pages = self._pages(s)
            want += max(0, (n + ops.KV_PAGE - 1) // ops.KV_PAGE - len(pages))
            want += int(n > self.lens[s] and bool(pages) and self.lens[s] % ops.KV_PAGE != 0 and self.pool.refs[pages[-1]] > 1)
        if want > len(self.pool.free):
            raise RuntimeError(f"paged KV pool exhausted: {want} pages wanted, {len(self.pool.free)} of {self.pool.npages - 1} free "
                               f"({ops.KV_PAGE} tokens each): raise pool_pages")
        for s, n in enumerate(need):
            pages = self._pages(s)
            # copy-on-write: the last, partially filled page is shared with a snapshot and this call appends to it
            if n > self.lens[s] and pages and self.lens[s] % ops.KV_PAGE and self.pool.refs[pages[-1]] > 1:
                old, new = pages[-1], self.pool.alloc()
                for sl in self.pool.slabs:
                    sl.k[new].copy_(sl.k[old])
                    sl.vt[new].copy_(sl.vt[old])
                self.pool.unref(old)
                pages[-1] = new
                changed.append((s, len(pages) - 1, new))
            while len(pages) * ops.KV_PAGE < n:
                pages.append(self.pool.alloc())
                changed.append((s, len(pages) - 1, pages[-1]))
        if changed:
            idx = torch.tensor([[self._seg0 + s, p] for s, p, _ in changed], dtype=torch.long)
            val = torch.tensor([v for _, _, v in changed], dtype=torch.int32)
            self.pool.table.index_put_((idx[:, 0].to(self.device), idx[:, 1].to(self.device)), val.to(self.device))

    def reserve(self, nseg, cap, nkv, hd, device):
        """NaiveCache.reserve pre-sizes slabs; here it only creates the pool and the table (pages are taken as contexts grow)"""
        self.ensure_tokens([0] * nseg, nkv, hd, device)

    def release(self, seg):
        """Return segment `seg`'s pages to the pool (its request is finished) and reset its length."""
        for p in self._pages(seg):
            self.pool.unref(p)
        self._pages(seg).clear()
        self.lens[seg] = 0

    def pages_in_use(self):
        return self.pool.npages - 1 - len(self.pool.free)

    def snapshot(self):
        """Logical copy sharing every page of the current contexts (reference counted); appends by either side to the shared, partially
        filled last page copy that one page first (ensure_tokens).  The snapshot has a table of its own over the same pools."""
        c = PagedCache(self._num_layers, self.pool_pages, self.max_pages * ops.KV_PAGE)
        c.lens = list(self.lens)
        c.nkv, c.hd, c.device = self.nkv, self.hd, self.device
        if self.pool is not None:
            src = self.pool
            c.pool = _PagePool.__new__(_PagePool)
            c.pool.table = torch.zeros_like(src.table[self._seg0:self._seg0 + len(self.lens)])
            c.pool.table.copy_(src.table[self._seg0:self._seg0 + len(self.lens)])
            c.pool.host_table = [list(self._pages(s)) for s in range(len(self.lens))]
            c.pool.free, c.pool.refs = src.free, src.refs            # ONE allocator: shared lists
            c.pool.npages, c.pool.nkv, c.pool.hd, c.pool.device = src.npages, src.nkv, src.hd, src.device
            c.pool.slabs = []
            for sl in src.slabs:
                v = ops.PagedSlab.__new__(ops.PagedSlab)
                v.k, v.vt, v.table, v.nkv, v.hd, v.cap = sl.k, sl.vt, c.pool.table, sl.nkv, sl.hd, sl.cap
                c.pool.slabs.append(v)
            for pages in c.pool.host_table:
                for p in pages:
                    src.refs[p] += 1
        return c

    def view_segments(self, start, end):
        """A cache over segments [start, end) of the same table and pools (no copy)."""
        v = PagedCache(self._num_layers, self.pool_pages, self.max_pages * ops.KV_PAGE)
        v.pool, v._seg0 = self.pool, self._seg0 + start
        v.lens = list(self.lens[start:end])
        v.nkv, v.hd, v.device = self.nkv, self.hd, self.device
        return v

    def packed_keys(self, layer):
        if self.pool is None or not any(self.lens):
            return None
        sl = self.pool.slabs[layer]
        out = []
        for s, n in enumerate(self.lens):
            pg = self._pages(s)
            for i in range((n + ops.KV_PAGE - 1) // ops.KV_PAGE):
                m = min(ops.KV_PAGE, n - i * ops.KV_PAGE)
                out.append(sl.k[pg[i], :, :m].transpose(0, 1))
        return torch.cat(out, 0)

    def packed_values(self, layer):
        if self.pool is None or not any(self.lens):
            return None
        sl = self.pool.slabs[layer]
        out = []
        for s, n in enumerate(self.lens):
            pg = self._pages(s)
            for i in range((n + ops.KV_PAGE - 1) // ops.KV_PAGE):
                m = min(ops.KV_PAGE, n - i * ops.KV_PAGE)
                out.append(sl.vt[pg[i], :, :, :m].permute(2, 0, 1))
        return torch.cat(out, 0)
