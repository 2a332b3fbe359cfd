"""VQA entry point - importable counterpart of the reference's notebook-style script
(codes/interactive_vqa_inferencer.py:58-336): same ``DEFAULT_CONFIG`` keys, same
``VQAInferencer(config).load_model()`` / ``.infer_single(image_path, prompt, temperature,
max_new_tokens, do_sample, show_image)`` returning the same result dict.  Keys that only
steered accelerate's CPU staging (enable_cpu_loading, offload_folder, max_mem_per_gpu,
enable_auto_bf16_conversion) are accepted and ignored: weights stream from the
memory-mapped safetensors file to the GPU tensor by tensor.
"""
import gc
import os
import time
from datetime import datetime
from typing import Any, Dict, Optional

import torch
from PIL import Image

from .bagel import Bagel
from . import packstore
from .checkpoint import checkpoint_getter, checkpoint_source_files
from .config import UniMedVLConfig
from .data_utils import add_special_tokens, pil_img2rgb
from .shapes import all_shapes
from .transforms import ImageTransform

DEFAULT_CONFIG = {
    "model_path": "/path/to/unimedvl_checkpoint",
    "target_gpu_device": "0",
    "max_mem_per_gpu": "40GiB",
    "temperature": 1.0,
    "max_new_tokens": 512,
    "do_sample": True,
    "seed": 42,
    "enable_cpu_loading": True,
    "enable_auto_bf16_conversion": True,
    "use_model_checkpoint": False,  # False = ema.safetensors, True = model.safetensors
    "offload_folder": "/tmp/bagel_offload",
}

# codes/data/default.yaml vlm_sft.image_transform_args (read by eval/vlm/utils.py:486-502)
VLM_SFT_TRANSFORM = dict(max_image_size=980, min_image_size=378, image_stride=14, max_pixels=2_007_040)


def build_transform():
    return ImageTransform(**VLM_SFT_TRANSFORM)


def process_conversation(images, conversation):
    return [pil_img2rgb(image) for image in images], conversation


def load_tokenizer(model_path):
    """Qwen2 byte-level BPE from the checkpoint's vocab.json / merges.txt / tokenizer_config.json
    (Qwen2Tokenizer.from_pretrained(model_path), interactive_vqa_inferencer.py:236): the in-tree implementation,
    id-for-id equal to the reference's class (tests/test_tokenizer_cpu.py); no transformers import at run time."""
    from .tokenizer import Qwen2Tokenizer
    return Qwen2Tokenizer.from_pretrained(model_path)


class VQAInferencer:
    def __init__(self, config: Optional[Dict[str, Any]] = None):
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.model = None
        self.tokenizer = None
        self.new_token_ids = None
        self.image_transform = None
        self.loaded = False

    def set_seed(self, seed):
        import random
        import numpy as np
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def convert_checkpoint_to_bf16(self, input_path, output_path):
        """reference method of the same name (one-time fp32 -> bf16 checkpoint conversion); not needed by load_model() here"""
        from .checkpoint import convert_checkpoint_to_bf16
        return convert_checkpoint_to_bf16(input_path, output_path)

    def load_model(self, model=None, tokenizer=None, new_token_ids=None):
        """Builds the engine from ``config['model_path']``.  Pre-built objects may be injected
        (tests / benches without a checkpoint)."""
        if self.loaded:
            print("Model already loaded")
            return
        self.set_seed(self.config["seed"])
        if model is None:
            model_path = self.config.get("model_path")
            if not model_path:
                raise ValueError("model_path required")
            cfg = UniMedVLConfig.from_checkpoint_dir(model_path)
            device = f"cuda:{self.config['target_gpu_device']}"
            # optional extras over the reference's config keys: "checkpoint_weight_path" overlays a fine-tuned checkpoint on the
            # base one (eval/vlm/utils.py:71-98); "llm_weight_dtype": "fp8" streams e4m3 LLM weights at decode
            cfg.llm_weight_dtype = self.config.get("llm_weight_dtype", "bf16")
            get = checkpoint_getter(model_path, all_shapes(cfg), self.config.get("checkpoint_weight_path"),
                                    self.config["use_model_checkpoint"])
            # fast path (not a reference key; the counterpart of its one-time ema_bf16.safetensors conversion,
            # interactive_vqa_inferencer.py:93-161): "packed_cache" (default True) keeps the device-ready weight images in
            # <model_path>/ema_packed_*.safetensors after the first load and reads them back on every later one
            t_load = time.time()
            store = packstore.attach(get, self.config.get("checkpoint_weight_path") or model_path, device, cfg,
                                     checkpoint_source_files(model_path, self.config.get("checkpoint_weight_path"),
                                                             self.config["use_model_checkpoint"]),
                                     enabled=bool(self.config.get("packed_cache", True)), extra_tag="_und")
            model = Bagel(cfg, get, device=device, visual_gen=False, visual_und=True)
            torch.cuda.synchronize()
            self.load_stats = {"load_s": round(time.time() - t_load, 3), "packed_cache": store.status, "from_packed": store.hits,
                               "built": store.misses}
            t_save = store.save()
            if t_save is not None:
                self.load_stats.update(packed_cache=store.status, packed_cache_write_s=round(t_save, 3))
            print(f"weights: {self.load_stats}")
            tokenizer = load_tokenizer(model_path)
            tokenizer, new_token_ids, _ = add_special_tokens(tokenizer)
        self.model, self.tokenizer, self.new_token_ids = model, tokenizer, new_token_ids
        # serving extra (not a reference key): every request prefills / decodes in ONE reserved cache, so the image span of a
        # request whose patch grid was seen before replays from a HIP graph (Bagel.forward_cache_update_vit); 0 = a fresh cache
        # per request as in the reference (bagel.py:1341)
        if hasattr(self.model, "chat_cache_tokens"):
            self.model.chat_cache_tokens = int(self.config.get("serving_cache_tokens", 8192))
        self.image_transform = build_transform()
        self.loaded = True
        self.show_gpu_memory()

    def show_gpu_memory(self):
        if torch.cuda.is_available():
            for i in range(torch.cuda.device_count()):
                allocated = torch.cuda.memory_allocated(i) / 1024 ** 3
                total = torch.cuda.get_device_properties(i).total_memory / 1024 ** 3
                print(f"GPU {i}: {allocated:.1f}GB / {total:.1f}GB ({allocated / total * 100:.1f}%)")

    def cleanup_memory(self):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        gc.collect()
        self.show_gpu_memory()

    def infer_single(self, image_path, prompt, temperature=None, max_new_tokens=None, do_sample=None, show_image=False):
        if not self.loaded:
            raise RuntimeError("Model not loaded, please call load_model() first")
        temperature = temperature if temperature is not None else self.config["temperature"]
        max_new_tokens = max_new_tokens if max_new_tokens is not None else self.config["max_new_tokens"]
        do_sample = do_sample if do_sample is not None else self.config["do_sample"]
        if isinstance(image_path, Image.Image):
            input_image, image_path = image_path.convert("RGB"), None
        else:
            if not os.path.exists(image_path):
                raise FileNotFoundError(f"Image file not found: {image_path}")
            input_image = Image.open(image_path).convert("RGB")
        start = time.time()
        images, conversation = process_conversation([input_image], prompt)
        answer = self.model.chat(self.tokenizer, self.new_token_ids, self.image_transform, images=images,
                                 prompt=conversation, max_length=max_new_tokens, do_sample=do_sample,
                                 temperature=temperature)
        return {"answer": answer, "input_image": input_image, "time": time.time() - start, "image_path": image_path,
                "prompt": prompt, "timestamp": datetime.now().strftime("%Y-%m-%d %H:%M:%S")}
