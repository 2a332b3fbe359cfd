"""On-disk fast path for the weight store: the device-ready images, written once, read back on every later load.

The reference converts its fp32 `ema.safetensors` to `ema_bf16.safetensors` the first time a model is loaded and reads
that file from then on (codes/interactive_vqa_inferencer.py:93-114,134-156; interactive_image_generator.py:97-161).  This
engine's load does more than a cast - every nn.Linear is re-tiled into MFMA fragment order, q/k/v and gate/up are fused,
and with llm_weight_dtype = "fp8" the e4m3 images and their power-of-two channel scales are derived - so what is kept is
the RESULT of all that: `ema_packed_<dtype>.safetensors` beside the checkpoint holds, per linear, the packed bf16 image
(+ bias, + e4m3 image, scales and the fp8-MFMA image when present) and every other tensor already in bf16.  A later load
reads each tensor straight onto the device and builds nothing; an fp32 source checkpoint is no longer read at all.

The file is only trusted for the checkpoint, the layout and the kernels it was made from: its metadata records
PACK_LAYOUT_VERSION, the sha256 of the packing kernels' source (csrc/pack.hip - not of the GEMM / attention / vision kernels, whose
edits do not change an image), the
weight / activation dtypes, (name, size, mtime) of every source file and the number of tensors / payload bytes written; on the
hit path every linear's (N, K) and every tensor's shape are checked against the model's shape table (shapes.all_shapes) and
every packed image against the size its (N, K) implies.  Anything else means "rebuild and overwrite".  Writing is best effort
(a read-only checkpoint directory just keeps loading the slow way), atomic (private temporary file, fsync, rename) and done by
local rank 0 only when several ranks load the same checkpoint."""
import json
import os
import time
import uuid

import torch

PACK_LAYOUT_VERSION = "umv-pack-1"     # P[n/16][k/32][lane][8] bf16; P8[n/16][k/64][lane][16 B] e4m3 + f32 pow2 scales; P8M fp8-MFMA image
_LIN_FIELDS = ("wp", "bias", "w8", "scale", "w8m")


def kernel_stamp():
    """sha256 of PACK_LAYOUT_VERSION + the sources that decide the bytes of a packed image: csrc/pack.hip and the headers it includes
    (common.h: the bf16 rounding helpers; gemm_internal.h: cvt_fp8x16 behind the deq / w8 images; the public header).  None when a
    source is not there - a cache whose maker cannot be identified is never trusted (and two unknowns never compare equal)"""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256(PACK_LAYOUT_VERSION.encode())
    try:
        for rel in ("csrc/pack.hip", "csrc/common.h", "csrc/gemm_internal.h", "../include/unimedvl_hip.h"):
            with open(os.path.join(here, rel), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()


def source_fingerprint(paths):
    out = []
    for p in paths:
        if p and os.path.exists(p):
            st = os.stat(p)
            out.append([os.path.basename(p), int(st.st_size), int(st.st_mtime_ns)])
    return json.dumps(out)


class PackStore:
    """`get.pack_store` of a checkpoint getter: weights.py asks it for every linear / tensor before building one."""

    def __init__(self, path, device, dtype_tag, source_files, enabled=True, expected_shapes=None):
        self.path, self.device, self.dtype_tag = path, torch.device(device), dtype_tag
        self.fingerprint = source_fingerprint(source_files)
        self.kstamp = kernel_stamp()
        self.expected = dict(expected_shapes or {})      # checkpoint tensor name -> shape (shapes.all_shapes)
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
                elif self.kstamp is None or md.get("kernel_stamp") != self.kstamp:
                    self.status = "made by other (or unidentifiable) packing kernels"
                else:
                    keys = set(f.keys())
                    with open(path, "rb") as fh:
                        header = int.from_bytes(fh.read(8), "little")
                    if int(md.get("n_tensors", -1)) != len(keys) or 8 + header + int(md.get("payload_bytes", -1)) != os.path.getsize(path):
                        self.status = "incomplete file (tensor count / payload size differ from what was written)"
                    else:
                        self.f, self.meta, self.status, self.keys = f, json.loads(md.get("linears", "{}")), "hit", keys
            except Exception as e:     # a truncated / foreign file: rebuild
                self.status = f"unreadable ({type(e).__name__}: {e})"

    @property
    def reading(self):
        return self.f is not None

    # ------------------------------------------------------------------ hit-path validation
    def _expected_nk(self, key):
        """(N, K) the model's shape table implies for packed linear `key` (fused q/k/v and gate/up keys add up their parts)"""
        es = self.expected
        if key.endswith(".weight") and key in es:
            parts = [key]
        elif "qkv_proj" in key:
            parts = [key.replace("qkv_proj", n + "_proj") + ".weight" for n in "qkv"]
        elif "gate_up_proj" in key:
            parts = [key.replace("gate_up_proj", n + "_proj") + ".weight" for n in ("gate", "up")]
        else:
            return None
        if not all(q in es for q in parts):
            return None
        K = 1
        for d in es[parts[0]][1:]:
            K *= int(d)
        return sum(int(es[q][0]) for q in parts), K

    def _check_linear(self, key, m, parts):
        from . import _lib
        nk = self._expected_nk(key)
        if nk is not None and (m["N"] != nk[0] or not (nk[1] <= m["K"] < nk[1] + 32)):
            raise _lib.UmvError(f"{self.path}: {key} is {m['N']} x {m['K']} in the packed file, the model wants {nk[0]} x {nk[1]}")
        if m["th"] == 16 and parts["wp"] is not None:
            want = _lib.load().umv_packed_weight_elems(m["N"], m["K"])
            if parts["wp"].numel() != want:
                raise _lib.UmvError(f"{self.path}: {key}: packed image of {parts['wp'].numel()} elements, {m['N']} x {m['K']} needs {want}")
        if parts["bias"] is not None and parts["bias"].numel() != m["N"]:
            raise _lib.UmvError(f"{self.path}: {key}: bias of {parts['bias'].numel()} for N = {m['N']}")

    # ------------------------------------------------------------------ linears
    def linear(self, key, build):
        """The PackedLinear `key`: from the packed file when it is there, else build() - and remember it for save()."""
        from . import ops
        if self.reading and key in self.meta:
            m = self.meta[key]
            t0 = time.time()
            parts = {fld: (self.f.get_tensor(f"{key}::{fld}") if f"{key}::{fld}" in self.keys else None) for fld in _LIN_FIELDS}
            self.t_read += time.time() - t0
            self._check_linear(key, m, parts)
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
            if name in self.expected:
                want = 1
                for d in self.expected[name]:
                    want *= int(d)
                if t.numel() != want:
                    from . import _lib
                    raise _lib.UmvError(f"{self.path}: {name} has {t.numel()} elements, the model wants {tuple(self.expected[name])}")
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
        # one writer per NODE (the ranks build identical files): LOCAL_RANK decides; a launcher that sets only RANK gives no node-local
        # information, so every rank writes (the private temporary name + rename keep that safe, and every node ends up with its file)
        if int(os.environ.get("LOCAL_RANK", "0") or 0) != 0:
            self.status = "not written (local rank 0 writes)"
            self.pending, self.pending_lin = {}, {}
            return None
        t0 = time.time()
        tmp = f"{self.path}.{os.getpid()}.{uuid.uuid4().hex}.tmp"      # private: another process may be writing the same file
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
            payload = sum(v.numel() * v.element_size() for v in tensors.values())
            save_file(tensors, tmp, metadata=dict(layout_version=PACK_LAYOUT_VERSION, dtype_tag=self.dtype_tag, source=self.fingerprint,
                                                  kernel_stamp=self.kstamp or "", n_tensors=str(len(tensors)), payload_bytes=str(payload),
                                                  linears=json.dumps(pending_meta)))
            fd = os.open(tmp, os.O_RDONLY)
            try:
                os.fsync(fd)          # the data is on disk before the name is
            finally:
                os.close(fd)
            os.replace(tmp, self.path)
            self.status = "written"
        except Exception as e:       # read-only directory, disk full: keep loading the slow way
            self.status = f"not written ({type(e).__name__}: {e})"
            try:
                os.remove(tmp)
            except OSError:
                pass
        self.pending, self.pending_lin = {}, {}
        return time.time() - t0


def attach(get, model_path, device, cfg, source_files, enabled=True, extra_tag=""):
    """Give the checkpoint getter `get` a PackStore for `<model_path>/ema_packed_<dtypes>.safetensors`."""
    from . import shapes
    tag = f"w-{cfg.llm_weight_dtype}_a-{cfg.llm_act_dtype}{extra_tag}"
    try:
        expected = shapes.all_shapes(cfg)
    except Exception:      # a config the shape table does not cover: the packed images are still self-checked
        expected = None
    store = PackStore(os.path.join(model_path, f"ema_packed_{tag}.safetensors"), device, tag, source_files, enabled=enabled,
                      expected_shapes=expected)
    try:
        get.pack_store = store
    except AttributeError:           # a plain function object accepts attributes; anything exotic just goes without
        pass
    return store
