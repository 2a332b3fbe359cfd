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


def dict_getter(sd):
    return lambda name: sd[name]


def vae_getter(path):
    """ae.safetensors; keys may carry a 'module.' prefix (autoencoder.py:356)."""
    return SafetensorsGetter(path, strip_prefix="module.")
