"""Model dimensions.  The reference reads them from the checkpoint's llm_config.json /
vit_config.json (interactive_vqa_inferencer.py:206-213) and hard-codes the VAE
(autoencoder.py:338-349); nothing here is baked into the kernels."""
import json
import os
from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass
class UniMedVLConfig:
    # Qwen2-MoT LLM
    hidden: int = 3584
    layers: int = 28
    heads: int = 28
    kv_heads: int = 4
    inter: int = 18944
    vocab: int = 152064
    rope_theta: float = 1e6
    rms_eps: float = 1e-6
    max_position: int = 32768
    # "bf16" (the reference's precision) or "fp8": weight-only e4m3 with power-of-two channel scales for the LLM
    # linear layers and lm_head (BASELINE.json configs[4]; include/unimedvl_hip.h umv_quantize_pack_weight_fp8)
    llm_weight_dtype: str = "bf16"
    # "fp8" (needs llm_weight_dtype == "fp8"): W8A8 - every LLM forward that is not a one-token decode step rounds the
    # activations of its linear layers per row through e4m3 and runs them on the fp8 matrix instruction
    # (umv_gemm_fp8a8w); decode steps keep bf16 activations on the e4m3 weight stream
    llm_act_dtype: str = "bf16"
    # SigLIP NaViT (the scripts drop the last layer: interactive_vqa_inferencer.py:213)
    vit_hidden: int = 1152
    vit_layers: int = 26
    vit_heads: int = 16
    vit_inter: int = 4304
    patch: int = 14
    vit_side: int = 70
    ln_eps: float = 1e-6
    # latent / VAE
    max_latent: int = 64
    latent_patch: int = 2
    z_channels: int = 16
    vae_ch: int = 128
    vae_mult: Tuple[int, ...] = (1, 2, 4, 4)
    vae_res: int = 2
    scale_factor: float = 0.3611
    shift_factor: float = 0.1159

    @property
    def head_dim(self):
        return self.hidden // self.heads

    @property
    def vit_head_dim(self):
        return self.vit_hidden // self.vit_heads

    @property
    def vae_downsample(self):
        return 2 ** (len(self.vae_mult) - 1)

    @property
    def latent_downsample(self):
        return self.vae_downsample * self.latent_patch

    @staticmethod
    def from_dict(d):
        keys = UniMedVLConfig.__dataclass_fields__.keys()
        kw = {k: d[k] for k in keys if k in d}
        if "vae_mult" in kw:
            kw["vae_mult"] = tuple(kw["vae_mult"])
        return UniMedVLConfig(**kw)

    @staticmethod
    def from_checkpoint_dir(path, max_latent_size=64, vit_max_num_patch_per_side=70):
        """llm_config.json / vit_config.json as shipped with the HF checkpoint."""
        llm = json.load(open(os.path.join(path, "llm_config.json")))
        vit = json.load(open(os.path.join(path, "vit_config.json")))
        return UniMedVLConfig(
            hidden=llm["hidden_size"], layers=llm["num_hidden_layers"], heads=llm["num_attention_heads"],
            kv_heads=llm["num_key_value_heads"], inter=llm["intermediate_size"], vocab=llm["vocab_size"],
            rope_theta=llm.get("rope_theta", 1e6), rms_eps=llm.get("rms_norm_eps", 1e-6),
            max_position=llm.get("max_position_embeddings", 32768),
            vit_hidden=vit["hidden_size"], vit_layers=vit["num_hidden_layers"] - 1,
            vit_heads=vit["num_attention_heads"], vit_inter=vit["intermediate_size"], patch=vit["patch_size"],
            vit_side=vit_max_num_patch_per_side, ln_eps=vit.get("layer_norm_eps", 1e-6), max_latent=max_latent_size)

    def to_dict(self):
        return asdict(self)
