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
