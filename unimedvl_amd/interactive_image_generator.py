"""Image generation / editing entry point - importable counterpart of the reference's
notebook-style script (codes/interactive_image_generator.py:56-275): same ``DEFAULT_CONFIG``
keys, ``ImageGenerator(config).load_model()`` building ``.inferencer`` (an
InterleaveInferencer), ``.set_seed``.  The example cell (understand-then-edit,
interactive_image_generator.py:290-397) is ``edit_with_understanding``.
"""
import os
import time
from typing import Any, Dict, Optional

import numpy as np
import torch

from .bagel import Bagel
from . import packstore
from .checkpoint import checkpoint_getter, checkpoint_source_files, vae_getter
from .config import UniMedVLConfig
from .data_utils import add_special_tokens
from .inferencer import InterleaveInferencer
from .shapes import all_shapes
from .transforms import ImageTransform
from .vae import AutoEncoder

DEFAULT_CONFIG = {
    "model_path": "/path/to/unimedvl_checkpoint",
    "target_gpu_device": "0",
    "max_mem_per_gpu": "40GiB",
    "enable_cpu_loading": True,
    "use_model_checkpoint": False,
    "enable_auto_bf16_conversion": True,
    "offload_folder": "/tmp/bagel_offload",
    "seed": 42,
    "vae_transform_size": (1024, 32, 16),
    "vit_transform_size": (980, 387, 14),
    "text_do_sample": False,
    "text_temperature": 0.3,
}


class ImageGenerator:
    def __init__(self, config: Optional[Dict[str, Any]] = None):
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.model = self.vae_model = self.tokenizer = None
        self.vae_transform = self.vit_transform = None
        self.new_token_ids = None
        self.inferencer = None
        self.loaded = False

    def set_seed(self, seed):
        import random
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def convert_checkpoint_to_bf16(self, input_path, output_path):
        """reference method of the same name (one-time fp32 -> bf16 checkpoint conversion); not needed by load_model() here"""
        from .checkpoint import convert_checkpoint_to_bf16
        return convert_checkpoint_to_bf16(input_path, output_path)

    def load_model(self, model=None, vae_model=None, tokenizer=None, new_token_ids=None):
        if self.loaded:
            print("Model already loaded")
            return
        self.set_seed(self.config["seed"])
        if model is None:
            model_path = self.config.get("model_path")
            if not model_path:
                raise ValueError("model_path required")
            cfg = UniMedVLConfig.from_checkpoint_dir(model_path, max_latent_size=64, vit_max_num_patch_per_side=70)
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
                                     enabled=bool(self.config.get("packed_cache", True)), extra_tag="_gen")
            model = Bagel(cfg, get, device=device, visual_gen=True, visual_und=True)
            torch.cuda.synchronize()
            self.load_stats = {"load_s": round(time.time() - t_load, 3), "packed_cache": store.status, "from_packed": store.hits,
                               "built": store.misses}
            t_save = store.save()
            if t_save is not None:
                self.load_stats.update(packed_cache=store.status, packed_cache_write_s=round(t_save, 3))
            print(f"weights: {self.load_stats}")
            vae_model = AutoEncoder(cfg, vae_getter(os.path.join(model_path, "ae.safetensors")), device=device)
            from .interactive_vqa_inferencer import load_tokenizer
            tokenizer, new_token_ids, _ = add_special_tokens(load_tokenizer(model_path))
        self.model, self.vae_model, self.tokenizer, self.new_token_ids = model, vae_model, tokenizer, new_token_ids
        self.vae_transform = ImageTransform(*self.config["vae_transform_size"])
        self.vit_transform = ImageTransform(*self.config["vit_transform_size"])
        self.inferencer = InterleaveInferencer(model=self.model, vae_model=self.vae_model, tokenizer=self.tokenizer,
                                               vae_transform=self.vae_transform, vit_transform=self.vit_transform,
                                               new_token_ids=self.new_token_ids)
        self.loaded = True

    def show_gpu_memory(self):
        if torch.cuda.is_available():
            for i in range(torch.cuda.device_count()):
                allocated = torch.cuda.memory_allocated(i) / 1024 ** 3
                total = torch.cuda.get_device_properties(i).total_memory / 1024 ** 3
                print(f"GPU {i}: {allocated:.1f}GB / {total:.1f}GB ({allocated / total * 100:.1f}%)")

    def edit_with_understanding(self, image, edit_instruction, use_thinking=False, seed=None, cfg_text_scale=4.0,
                                cfg_img_scale=2.0, cfg_interval=(0.0, 1.0), timestep_shift=3.0, num_timesteps=50,
                                cfg_renorm_min=0.0, cfg_renorm_type="text_channel", max_think_token_n=1024):
        """The script's cell 4: describe the image, then edit it conditioned on image + text
        (interactive_image_generator.py:290-397)."""
        if not self.loaded:
            raise RuntimeError("Model not loaded, please call load_model() first")
        if seed is not None:
            self.set_seed(seed)
        understanding = self.inferencer(image=image, text=edit_instruction, understanding_output=True,
                                        think=use_thinking, do_sample=self.config["text_do_sample"],
                                        text_temperature=self.config["text_temperature"],
                                        max_think_token_n=max_think_token_n)
        h, w = self.inferencer._calculate_target_size_with_aspect_ratio(*image.size)
        edited = self.inferencer(image=image, text=edit_instruction, think=use_thinking, cfg_text_scale=cfg_text_scale,
                                 cfg_img_scale=cfg_img_scale, cfg_interval=list(cfg_interval), timestep_shift=timestep_shift,
                                 num_timesteps=num_timesteps, cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type,
                                 image_shapes=(h, w), do_sample=self.config["text_do_sample"],
                                 text_temperature=self.config["text_temperature"])
        return {"understanding": understanding["text"], "image": edited["image"], "text": edited["text"]}
