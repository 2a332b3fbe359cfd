"""On-disk fast path for the weight store: the device-ready images, written once, read back on every later load.

The reference converts its fp32 `ema.safetensors` to `ema_bf16.safetensors` the first time a model is loaded and reads
that file from then on (codes/interactive_vqa_inferencer.py:93-114,134-156; interactive_image_generator.py:97-161).  This
engine's load does more than a cast - every nn.Linear is re-tiled into MFMA fragment order, q/k/v and gate/up are fused,
and with llm_weight_dtype = "fp8" the e4m3 images and their power-of-two channel scales are derived - so what is kept is
the RESULT of all that: `ema_packed_<dtype>.safetensors` beside the checkpoint holds, per linear, the packed bf16 image
(+ bias, + e4m3 image, scales and the fp8-MFMA image when present) and every other tensor already in bf16.  A later load
reads each tensor straight onto the device and builds nothing; an fp32 source checkpoint is no longer read at all.

The file is only trusted for the checkpoint and the layout it was made from: its metadata records PACK_LAYOUT_VERSION (bump
it whenever a packing kernel changes its image), the weight / activation dtypes and (name, size, mtime) of every source
file; anything else means "rebuild and overwrite".  Writing is best effort (a read-only checkpoint directory just keeps
loading the slow way)."""
import json
import os
import time

import torch

PACK_LAYOUT_VERSION = "umv-pack-1"     # P[n/16][k/32][lane][8] bf16; P8[n/16][k/64][lane][16 B] e4m3 + f32 pow2 scales; P8M fp8-MFMA image
_LIN_FIELDS = ("wp", "bias", "w8", "scale", "w8m")


def source_fingerprint(paths):
    out = []
    for p in paths:
        if p and os.path.exists(p):
            st = os.stat(p)
            out.append([os.path.basename(p), int(st.st_size), int(st.st_mtime_ns)])
    return json.dumps(out)


class PackStore:
    """`get.pack_store` of a checkpoint getter: weights.py asks it for every linear / tensor before building one."""

    def __init__(self, path, device, dtype_tag, source_files, enabled=True):
        self.path, self.device, self.dtype_tag = path, torch.device(device), dtype_tag
        self.fingerprint = source_fingerprint(source_files)
        self.f = None            # open packed file (hit path)
        self.meta = {}
        self.pending = {}        # plain tensors to write (miss path)
        self.pending_lin = {}    # PackedLinear objects to write
        self.hits = self.misses = 0
        self.t_read = 0.0
        self.status = "disabled"
        if not enabled:
            return
        self.status = "absent"
        if os.path.exists(path):
            try:
                from safetensors import safe_open
                f = safe_open(path, framework="pt", device=str(self.device))
                md = f.metadata() or {}
                if md.get("layout_version") != PACK_LAYOUT_VERSION:
                    self.status = f"stale layout ({md.get('layout_version')} != {PACK_LAYOUT_VERSION})"
                elif md.get("dtype_tag") != dtype_tag:
                    self.status = f"other dtypes ({md.get('dtype_tag')} != {dtype_tag})"
                elif md.get("source") != self.fingerprint:
                    self.status = "made from other checkpoint files"
                else:
                    self.f, self.meta, self.status = f, json.loads(md.get("linears", "{}")), "hit"
                    self.keys = set(f.keys())
            except Exception as e:     # a truncated / foreign file: rebuild
                self.status = f"unreadable ({type(e).__name__}: {e})"

    @property
    def reading(self):
        return self.f is not None

    # ------------------------------------------------------------------ linears
    def linear(self, key, build):
        """The PackedLinear `key`: from the packed file when it is there, else build() - and remember it for save()."""
        from . import ops
        if self.reading and key in self.meta:
            m = self.meta[key]
            t0 = time.time()
            parts = {fld: (self.f.get_tensor(f"{key}::{fld}") if f"{key}::{fld}" in self.keys else None) for fld in _LIN_FIELDS}
            self.t_read += time.time() - t0
            lin = ops.PackedLinear(parts["wp"], parts["bias"], m["N"], m["K"], swiglu=m["swiglu"], th=m["th"], w8=parts["w8"],
                                   scale=parts["scale"])
            lin.w8m = parts["w8m"]
            self.hits += 1
            return lin
        lin = build()
        self.misses += 1
        if self.status != "disabled" and not self.reading:
            self.pending_lin[key] = lin          # its FINAL state is written (enable_fp8_mfma may still swap images)
        return lin

    def tensor(self, name, build):
        """A plain bf16 tensor (embeddings, norm gains, position tables)."""
        if self.reading and name in self.keys:
            t0 = time.time()
            t = self.f.get_tensor(name)
            self.t_read += time.time() - t0
            self.hits += 1
            return t
        t = build()
        self.misses += 1
        if self.status != "disabled" and not self.reading:
            self.pending[name] = t
        return t

    # ------------------------------------------------------------------ write
    def save(self):
        """Write what the miss path collected (no-op on the hit path / when disabled).  Returns the seconds spent, or None."""
        if self.reading or self.status == "disabled" or not (self.pending or self.pending_lin):
            return None
        t0 = time.time()
        try:
            from safetensors.torch import save_file
            torch.cuda.synchronize()
            pending_meta = {}
            for key, lin in self.pending_lin.items():
                pending_meta[key] = dict(N=lin.N, K=lin.K, swiglu=bool(lin.swiglu), th=lin.th)
                for fld in _LIN_FIELDS:
                    t = getattr(lin, fld, None)
                    if t is not None:
                        self.pending[f"{key}::{fld}"] = t
            tensors = {k: v.detach().to("cpu").contiguous() for k, v in self.pending.items()}
            tmp = self.path + ".tmp"
            save_file(tensors, tmp, metadata=dict(layout_version=PACK_LAYOUT_VERSION, dtype_tag=self.dtype_tag, source=self.fingerprint,
                                                  linears=json.dumps(pending_meta)))
            os.replace(tmp, self.path)
            self.status = "written"
        except Exception as e:       # read-only directory, disk full: keep loading the slow way
            self.status = f"not written ({type(e).__name__}: {e})"
            try:
                os.remove(self.path + ".tmp")
            except OSError:
                pass
        self.pending, self.pending_lin = {}, {}
        return time.time() - t0


def attach(get, model_path, device, cfg, source_files, enabled=True, extra_tag=""):
    """Give the checkpoint getter `get` a PackStore for `<model_path>/ema_packed_<dtypes>.safetensors`."""
    tag = f"w-{cfg.llm_weight_dtype}_a-{cfg.llm_act_dtype}{extra_tag}"
    store = PackStore(os.path.join(model_path, f"ema_packed_{tag}.safetensors"), device, tag, source_files, enabled=enabled)
    try:
        get.pack_store = store
    except AttributeError:           # a plain function object accepts attributes; anything exotic just goes without
        pass
    return store
