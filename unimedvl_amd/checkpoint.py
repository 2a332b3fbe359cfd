"""Checkpoint access: safetensors -> ``get(name)`` callables for the weight store.

Replaces the accelerate-based loaders of the reference entry points
(codes/interactive_vqa_inferencer.py:93-189, interactive_image_generator.py:97-195,
codes/modeling/autoencoder.py:337-360): tensors are read one at a time straight from the
memory-mapped safetensors file (``ema_bf16.safetensors`` if present, else ``ema.safetensors``
/ ``model.safetensors``), cast to bf16 on the way to the device and re-tiled there; no
CPU-side full copy, no auto-converted second file on disk.
"""
import os

import torch


def find_weights_file(model_path, use_model_checkpoint=False):
    base = "model" if use_model_checkpoint else "ema"
    for name in (f"{base}_bf16.safetensors", f"{base}.safetensors"):
        p = os.path.join(model_path, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no {base}(_bf16).safetensors under {model_path}")


def convert_checkpoint_to_bf16(input_path, output_path):
    """The reference's one-time conversion utility (interactive_vqa_inferencer.py:93-114 / interactive_image_generator.py:97-118),
    kept for callers that use it directly: write `output_path` = the tensors of `input_path` in bf16.  False when the input does
    not exist, True otherwise (an input that already is bf16 - judged by its first tensor, as the reference does - is copied).
    The engine itself does not need the file: it casts on the way to the device and keeps its own packed fast-path file."""
    import shutil
    if not os.path.exists(input_path):
        return False
    from safetensors import safe_open
    from safetensors.torch import save_file
    with safe_open(input_path, framework="pt", device="cpu") as f:
        names = list(f.keys())
        if names and f.get_tensor(names[0]).dtype == torch.bfloat16:
            if input_path != output_path:
                shutil.copy(input_path, output_path)
            return True
        out = {k: f.get_tensor(k).to(torch.bfloat16) for k in names}     # every tensor, as the reference (tensor.to(bfloat16))
    save_file(out, output_path)
    return True


class SafetensorsGetter:
    """get(name) over one safetensors file; validates shapes against the expected table."""

    def __init__(self, path, expected_shapes=None, strip_prefix=""):
        from safetensors import safe_open
        self.f = safe_open(path, framework="pt", device="cpu")
        self.keys = set(self.f.keys())
        self.expected = expected_shapes
        self.strip = strip_prefix

    def __call__(self, name):
        key = name
        if key not in self.keys and self.strip + name in self.keys:
            key = self.strip + name
        if key not in self.keys:
            raise KeyError(f"tensor '{name}' missing from the checkpoint")
        t = self.f.get_tensor(key)
        if self.expected is not None and name in self.expected and tuple(t.shape) != tuple(self.expected[name]):
            raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != expected {tuple(self.expected[name])}")
        return t


class OverlayGetter:
    """Fine-tuned checkpoint over a base checkpoint: a tensor comes from the first getter that has it.  Replaces the
    two strict=False load_state_dict passes of the evaluation loader (codes/eval/vlm/utils.py:71-98: base
    `ema.safetensors` first "to ensure no missing weights", then the fine-tuned `ema(_bf16).safetensors` on top)."""

    def __init__(self, *getters):
        self.getters = [g for g in getters if g is not None]
        if not self.getters:
            raise ValueError("OverlayGetter needs at least one checkpoint")

    def __call__(self, name):
        err = None
        for g in self.getters:
            try:
                return g(name)
            except KeyError as e:
                err = e
        raise err


def checkpoint_getter(model_path, expected_shapes=None, checkpoint_weight_path=None, use_model_checkpoint=False):
    """get(name) for a checkpoint directory; with `checkpoint_weight_path` the fine-tuned ema(_bf16).safetensors found
    there overlays the base file of `model_path` (eval/vlm/utils.py:71-98)."""
    base = SafetensorsGetter(find_weights_file(model_path, use_model_checkpoint), expected_shapes)
    if checkpoint_weight_path is None:
        return base
    return OverlayGetter(SafetensorsGetter(find_weights_file(checkpoint_weight_path, False), expected_shapes), base)


def checkpoint_source_files(model_path, checkpoint_weight_path=None, use_model_checkpoint=False):
    """the safetensors files checkpoint_getter reads (the packed fast-path file is keyed on their names / sizes / mtimes)"""
    files = [find_weights_file(model_path, use_model_checkpoint)]
    if checkpoint_weight_path is not None:
        files.insert(0, find_weights_file(checkpoint_weight_path, False))
    return files


def dict_getter(sd):
    return lambda name: sd[name]


def vae_getter(path):
    """ae.safetensors; keys may carry a 'module.' prefix (autoencoder.py:356)."""
    return SafetensorsGetter(path, strip_prefix="module.")
