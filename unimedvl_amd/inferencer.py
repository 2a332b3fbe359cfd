"""Interleaved text/image orchestration - drop-in for the reference's
``InterleaveInferencer`` (codes/inferencer.py:31-680): same constructor, same methods,
same keyword arguments and defaults, same return shapes.

What changes underneath:
  * the reference's ``deepcopy(gen_context)`` per text item and in ``gen_text`` (inferencer.py:261,587,600,607) is a
    whole-KV copy; here the three CFG contexts are PREFIX SNAPSHOTS of one in-place cache (kvcache.NaiveCache.snapshot:
    shared slabs + frozen lengths, copy-on-write), ``cfg_img_context`` gets a prefill of its own only after an image
    has made its token sequence differ from ``gen_context``'s, and ``gen_text`` decodes in place beyond the committed
    length and then puts the length back - the caller's context is never advanced by decoding, as in the reference;
  * in understanding mode the reference still prefills the never-used ``cfg_img_context``
    for every text item (inferencer.py:602); that work is skipped when
    ``understanding_output=True`` because nothing ever reads it;
  * all compute runs on the MI355X engine (unimedvl_amd.Bagel).
"""
from typing import Any, Dict, List, Optional, Union

import torch
from PIL import Image

from .data_utils import pil_img2rgb
from .kvcache import NaiveCache

VLM_THINK_SYSTEM_PROMPT = '''You should first think about the reasoning process in the mind and then provide the user with the answer.
The reasoning process is enclosed within <think> </think> tags, i.e. <think> reasoning process here </think> answer here'''

GEN_THINK_SYSTEM_PROMPT = '''You should first think about the planning process in your mind, and then generate the image.
The planning process is enclosed within <think> </think> tags; that is, <think> planning process here </think> image here.
'''


class InterleaveInferencer:
    def __init__(self, model, vae_model, tokenizer, vae_transform, vit_transform, new_token_ids):
        self.model = model
        self.vae_model = vae_model
        self.tokenizer = tokenizer
        self.vae_transform = vae_transform
        self.vit_transform = vit_transform
        self.new_token_ids = new_token_ids

    # ------------------------------------------------------------------ sizes
    def _calculate_target_size_with_aspect_ratio(self, original_width, original_height):
        """(height, width) the VAE transform would give this image (inferencer.py:42-71)."""
        rt = self.vae_transform.resize_transform
        max_size, min_size, stride, max_pixels = rt.max_size, rt.min_size, rt.stride, rt.max_pixels

        def make_divisible(value):
            return max(stride, int(round(value / stride) * stride))

        def apply_scale(width, height, scale):
            return make_divisible(round(width * scale)), make_divisible(round(height * scale))
        scale = min(max_size / max(original_width, original_height), 1.0)
        scale = max(scale, min_size / min(original_width, original_height))
        new_width, new_height = apply_scale(original_width, original_height, scale)
        if new_width * new_height > max_pixels:
            new_width, new_height = apply_scale(new_width, new_height, max_pixels / (new_width * new_height))
        if max(new_width, new_height) > max_size:
            new_width, new_height = apply_scale(new_width, new_height, max_size / max(new_width, new_height))
        return new_height, new_width

    # ------------------------------------------------------------------ contexts
    def init_gen_context(self):
        return {"kv_lens": [0], "ropes": [0],
                "past_key_values": NaiveCache(self.model.config.llm_config.num_hidden_layers)}

    @staticmethod
    def _snap(ctx):
        """What the reference gets from ``deepcopy(gen_context)`` (inferencer.py:261,587,600,607), without copying the KV:
        the snapshot shares the slabs up to the current lengths and copies itself only if IT is written to later
        (kvcache.NaiveCache.snapshot).  The source context may keep appending."""
        return {"kv_lens": list(ctx["kv_lens"]), "ropes": list(ctx["ropes"]), "past_key_values": ctx["past_key_values"].snapshot()}

    @torch.no_grad()
    def update_context_text(self, text, gen_context):
        gi, kv_lens, ropes = self.model.prepare_prompts(
            curr_kvlens=gen_context["kv_lens"], curr_rope=gen_context["ropes"], prompts=[text],
            tokenizer=self.tokenizer, new_token_ids=self.new_token_ids)
        pkv = self.model.forward_cache_update_text(gen_context["past_key_values"], **gi)
        gen_context["kv_lens"], gen_context["ropes"], gen_context["past_key_values"] = kv_lens, ropes, pkv
        return gen_context

    @torch.no_grad()
    def update_context_image(self, image, gen_context, vae=True, vit=True):
        assert vae or vit
        pkv, kv_lens, ropes = gen_context["past_key_values"], gen_context["kv_lens"], gen_context["ropes"]
        if vae:
            gi, kv_lens, ropes = self.model.prepare_vae_images(
                curr_kvlens=kv_lens, curr_rope=ropes, images=[image], transforms=self.vae_transform,
                new_token_ids=self.new_token_ids)
            pkv = self.model.forward_cache_update_vae(self.vae_model, pkv, **gi)
        if vit:
            gi, kv_lens, ropes = self.model.prepare_vit_images(
                curr_kvlens=kv_lens, curr_rope=ropes, images=[image], transforms=self.vit_transform,
                new_token_ids=self.new_token_ids)
            pkv = self.model.forward_cache_update_vit(pkv, **gi)
        gen_context["kv_lens"], gen_context["ropes"], gen_context["past_key_values"] = kv_lens, ropes, pkv
        return gen_context

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def gen_image(self, image_shape, gen_context, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_text_precontext=None,
                  cfg_img_precontext=None, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
                  num_timesteps=50, timestep_shift=3.0):
        m = self.model
        gi = m.prepare_vae_latent(curr_kvlens=gen_context["kv_lens"], curr_rope=gen_context["ropes"],
                                  image_sizes=[image_shape], new_token_ids=self.new_token_ids)
        gt = m.prepare_vae_latent_cfg(curr_kvlens=cfg_text_precontext["kv_lens"], curr_rope=cfg_text_precontext["ropes"],
                                      image_sizes=[image_shape])
        gim = m.prepare_vae_latent_cfg(curr_kvlens=cfg_img_precontext["kv_lens"], curr_rope=cfg_img_precontext["ropes"],
                                       image_sizes=[image_shape])
        unpacked_latent = m.generate_image(
            past_key_values=gen_context["past_key_values"],
            cfg_text_past_key_values=cfg_text_precontext["past_key_values"],
            cfg_img_past_key_values=cfg_img_precontext["past_key_values"],
            num_timesteps=num_timesteps, cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale,
            cfg_interval=cfg_interval, cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type,
            timestep_shift=timestep_shift, **gi,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
            cfg_img_packed_position_ids=gim["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gim["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=gim["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gim["cfg_packed_key_value_indexes"])
        return self.decode_image(unpacked_latent[0], image_shape)

    def decode_image(self, latent, image_shape):
        """latent tokens [h*w, p*p*c] -> PIL (inferencer.py:234-256): unpatchify, VAE decode,
        (x*0.5+0.5).clamp(0,1)*255 truncated to uint8."""
        pixels = self.vae_model.decode_tokens_to_uint8(latent, image_shape, self.model.latent_downsample,
                                                       self.model.latent_patch_size)
        return Image.fromarray(pixels.cpu().numpy())

    @torch.no_grad()
    def gen_text(self, gen_context, max_length: int = 500, do_sample: bool = True, temperature: float = 1.0):
        # the reference decodes on a deepcopy so that the context keeps its length (inferencer.py:261); here the decode
        # appends in place beyond the committed length and the length is simply put back afterwards
        cache = gen_context["past_key_values"]
        gi = self.model.prepare_start_tokens(gen_context["kv_lens"], gen_context["ropes"], self.new_token_ids)
        lens0 = list(cache.lens)
        try:
            ids = self.model.generate_text(past_key_values=cache, max_length=max_length, do_sample=do_sample,
                                           temperature=temperature, end_token_id=self.new_token_ids["eos_token_id"], **gi)
        finally:
            if cache.slabs is not None:
                cache.lens = lens0
        output = self.tokenizer.decode(ids[:, 0].cpu())
        return output.split("<|im_end|>")[0].split("<|im_start|>")[1]

    # ------------------------------------------------------------------ pipelines
    @torch.no_grad()
    def interleave_inference(self, input_lists: List[Union[str, Image.Image]], think=False, understanding_output=False,
                             max_think_token_n=1000, do_sample=False, text_temperature=0.3, cfg_text_scale=3.0,
                             cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), timestep_shift=3.0, num_timesteps=50,
                             cfg_renorm_min=0.0, cfg_renorm_type="global", image_shapes=(1024, 1024)
                             ) -> List[Union[str, Image.Image]]:
        """inferencer.py:552-638."""
        output_list = []
        need_cfg = not understanding_output
        gen_context = self.init_gen_context()
        # The three contexts of inferencer.py:587-607 share prefixes: cfg_text is gen_context as it was before the latest
        # text item, cfg_img is gen_context without the images.  The reference deep-copies / re-prefills them; here they are
        # snapshots of gen_context's in-place cache, and cfg_img only gets a prefill of its own once an image has made
        # the two token sequences differ (until then it IS gen_context's prefix: same tokens, same positions).
        cfg_text_context = self._snap(gen_context)
        cfg_img_context = self._snap(gen_context)
        img_in_sync = True            # cfg_img_context holds exactly gen_context's tokens so far
        if think:
            system_prompt = VLM_THINK_SYSTEM_PROMPT if understanding_output else GEN_THINK_SYSTEM_PROMPT
            gen_context = self.update_context_text(system_prompt, gen_context)
            if need_cfg:
                cfg_img_context = self._snap(gen_context)
        for input_term in input_lists:
            if isinstance(input_term, str):
                if need_cfg:
                    cfg_text_context = self._snap(gen_context)
                gen_context = self.update_context_text(input_term, gen_context)
                if need_cfg:
                    cfg_img_context = (self._snap(gen_context) if img_in_sync
                                       else self.update_context_text(input_term, cfg_img_context))
            elif isinstance(input_term, Image.Image):
                input_term = self.vae_transform.resize_transform(pil_img2rgb(input_term))
                gen_context = self.update_context_image(input_term, gen_context, vae=not understanding_output)
                img_in_sync = False
                if need_cfg:
                    cfg_text_context = self._snap(gen_context)
            else:
                raise ValueError(f"Unsupported input type: {type(input_term)}")
        if understanding_output:
            output_list.append(self.gen_text(gen_context, do_sample=do_sample, temperature=text_temperature,
                                             max_length=max_think_token_n))
        else:
            if think:
                gen_text = self.gen_text(gen_context, do_sample=do_sample, temperature=text_temperature,
                                         max_length=max_think_token_n)
                gen_context = self.update_context_text(gen_text, gen_context)
                output_list.append(gen_text)
            output_list.append(self.gen_image(
                image_shapes, gen_context, cfg_text_precontext=cfg_text_context, cfg_img_precontext=cfg_img_context,
                cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval,
                timestep_shift=timestep_shift, num_timesteps=num_timesteps, cfg_renorm_min=cfg_renorm_min,
                cfg_renorm_type=cfg_renorm_type))
        return output_list

    @torch.no_grad()
    def interleave_inference_for_vqa_reconstruction_ver1(
            self, input_lists: List[Union[str, Image.Image]], reconstruct_image: bool = False, think: bool = False,
            understanding_output: bool = True, max_think_token_n: int = 1000, do_sample: bool = False,
            text_temperature: float = 0.3, cfg_text_scale: float = 3.0, cfg_img_scale: float = 1.5,
            cfg_interval: list = (0.4, 1.0), timestep_shift: float = 3.0, num_timesteps: int = 50,
            cfg_renorm_min: float = 0.0, cfg_renorm_type: str = "global", image_shapes: tuple = (1024, 1024)
    ) -> List[Union[str, Image.Image]]:
        """VQA, then optionally reconstruct every input image from the answer (inferencer.py:282-362)."""
        output_list = []
        vqa_context = self.init_gen_context()
        vqa_img_context = self._snap(vqa_context)
        for input_term in input_lists:
            if isinstance(input_term, str):
                vqa_context = self.update_context_text(input_term, vqa_context)
                vqa_img_context = self.update_context_text(input_term, vqa_img_context)
            elif isinstance(input_term, Image.Image):
                processed = self.vae_transform.resize_transform(pil_img2rgb(input_term))
                vqa_context = self.update_context_image(processed, vqa_context, vae=True, vit=True)
            else:
                raise ValueError(f"Unsupported input type: {type(input_term)}")
        vqa_answer = self.gen_text(vqa_context, do_sample=do_sample, temperature=text_temperature,
                                   max_length=max_think_token_n)
        output_list.append(vqa_answer)
        if not reconstruct_image or not vqa_answer or not vqa_answer.strip():
            return output_list
        input_images = [item for item in input_lists if isinstance(item, Image.Image)]
        if not input_images:
            return output_list
        cfg_text_precontext = self._snap(vqa_context)               # [images, question]: a prefix of full_context
        cfg_img_precontext = self.update_context_text(vqa_answer, vqa_img_context)
        full_context = self.update_context_text(vqa_answer, vqa_context)   # appended in place (vqa_context is not used again)
        for original_image in input_images:
            w, h = original_image.size
            target = self._calculate_target_size_with_aspect_ratio(w, h)
            generated = self.gen_image(
                target, full_context, cfg_text_precontext=cfg_text_precontext, cfg_img_precontext=cfg_img_precontext,
                cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval,
                cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type, num_timesteps=num_timesteps,
                timestep_shift=timestep_shift)
            output_list.append(generated)
            processed = self.vae_transform.resize_transform(pil_img2rgb(generated))
            full_context = self.update_context_image(processed, full_context, vae=True, vit=False)
            cfg_text_precontext = self.update_context_image(processed, cfg_text_precontext, vae=True, vit=False)
        return output_list

    def _vqa_then_rebuild(self, input_lists, reconstruct_image, first_only, do_sample, text_temperature, max_think_token_n,
                          cfg_interval, timestep_shift, num_timesteps, cfg_renorm_min, cfg_renorm_type):
        """Shared body of the two older VQA+reconstruction variants (inferencer.py:366-549): answer the question with
        the image(s) in both ViT and VAE form, then regenerate the image(s) from a FRESH context = [image, answer], with
        cfg_text context = [image] and cfg_img context = [answer], both guidance scales fixed at 7.0 by the reference."""
        ctx = self.init_gen_context()
        for item in input_lists:
            if isinstance(item, str):
                ctx = self.update_context_text(item, ctx)
            elif isinstance(item, Image.Image):
                ctx = self.update_context_image(self.vae_transform.resize_transform(pil_img2rgb(item)), ctx, vae=True, vit=True)
            else:
                raise ValueError(f"Unsupported input type: {type(item)}")
        answer = self.gen_text(ctx, do_sample=do_sample, temperature=text_temperature, max_length=max_think_token_n)
        outputs = [answer]
        pictures = [it for it in input_lists if isinstance(it, Image.Image)]
        if not reconstruct_image or not answer or not answer.strip() or not pictures:
            return outputs
        for picture in (pictures[:1] if first_only else pictures):
            w, h = picture.size
            target = self._calculate_target_size_with_aspect_ratio(w, h)
            resized = self.vae_transform.resize_transform(pil_img2rgb(picture))
            image_and_answer = self.update_context_image(resized, self.init_gen_context(), vae=True, vit=True)
            image_only = self._snap(image_and_answer)           # the cfg_text context is the prefix before the answer
            image_and_answer = self.update_context_text(answer, image_and_answer)
            answer_only = self.update_context_text(answer, self.init_gen_context())
            outputs.append(self.gen_image(
                target, image_and_answer, cfg_text_precontext=image_only, cfg_img_precontext=answer_only, cfg_text_scale=7.0,
                cfg_img_scale=7.0, cfg_interval=cfg_interval, timestep_shift=timestep_shift, num_timesteps=num_timesteps,
                cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type))
        return outputs

    @torch.no_grad()
    def interleave_inference_for_vqa_reconstruction_ver0_1(
            self, input_lists: List[Union[str, Image.Image]], reconstruct_image: bool = False, think: bool = False,
            understanding_output: bool = True, max_think_token_n: int = 1000, do_sample: bool = False,
            text_temperature: float = 0.3, cfg_text_scale: float = 3.0, cfg_img_scale: float = 1.5,
            cfg_interval: list = (0.4, 1.0), timestep_shift: float = 3.0, num_timesteps: int = 50,
            cfg_renorm_min: float = 0.0, cfg_renorm_type: str = "global", image_shapes: tuple = (1024, 1024)
    ) -> List[Union[str, Image.Image]]:
        """inferencer.py:366-463: VQA, then one reconstruction per input image.  (cfg_*_scale / think / image_shapes are
        accepted and ignored, as in the reference.)"""
        return self._vqa_then_rebuild(input_lists, reconstruct_image, False, do_sample, text_temperature, max_think_token_n,
                                      cfg_interval, timestep_shift, num_timesteps, cfg_renorm_min, cfg_renorm_type)

    @torch.no_grad()
    def interleave_inference_for_vqa_reconstruction_ver0(
            self, input_lists: List[Union[str, Image.Image]], reconstruct_image: bool = False, think: bool = False,
            understanding_output: bool = True, max_think_token_n: int = 1000, do_sample: bool = False,
            text_temperature: float = 0.3, cfg_text_scale: float = 3.0, cfg_img_scale: float = 1.5,
            cfg_interval: list = (0.4, 1.0), timestep_shift: float = 3.0, num_timesteps: int = 50,
            cfg_renorm_min: float = 0.0, cfg_renorm_type: str = "global", image_shapes: tuple = (1024, 1024)
    ) -> List[Union[str, Image.Image]]:
        """inferencer.py:466-549: VQA, then a reconstruction of the FIRST input image only."""
        return self._vqa_then_rebuild(input_lists, reconstruct_image, True, do_sample, text_temperature, max_think_token_n,
                                      cfg_interval, timestep_shift, num_timesteps, cfg_renorm_min, cfg_renorm_type)

    # ------------------------------------------------------------------ batch extension (additive)
    # The reference runs one sample per call (contexts are lists of length 1).  Samples are independent segments of
    # the packed NaViT sequence, so B samples with the same item structure (e.g. [image, question]) share every
    # forward: one ViT / prefill / decode / flow pass over B segments, per-sample EOS.
    def _batch_context(self, n):
        return {"kv_lens": [0] * n, "ropes": [0] * n,
                "past_key_values": NaiveCache(self.model.config.llm_config.num_hidden_layers)}

    def _update_batch_text(self, texts, ctx):
        gi, kv_lens, ropes = self.model.prepare_prompts(curr_kvlens=ctx["kv_lens"], curr_rope=ctx["ropes"], prompts=list(texts),
                                                        tokenizer=self.tokenizer, new_token_ids=self.new_token_ids)
        pkv = self.model.forward_cache_update_text(ctx["past_key_values"], **gi)
        ctx["kv_lens"], ctx["ropes"], ctx["past_key_values"] = kv_lens, ropes, pkv
        return ctx

    def _update_batch_image(self, images, ctx, vae=True, vit=True):
        pkv, kv_lens, ropes = ctx["past_key_values"], ctx["kv_lens"], ctx["ropes"]
        if vae:
            gi, kv_lens, ropes = self.model.prepare_vae_images(curr_kvlens=kv_lens, curr_rope=ropes, images=list(images),
                                                               transforms=self.vae_transform, new_token_ids=self.new_token_ids)
            pkv = self.model.forward_cache_update_vae(self.vae_model, pkv, **gi)
        if vit:
            gi, kv_lens, ropes = self.model.prepare_vit_images(curr_kvlens=kv_lens, curr_rope=ropes, images=list(images),
                                                               transforms=self.vit_transform, new_token_ids=self.new_token_ids)
            pkv = self.model.forward_cache_update_vit(pkv, **gi)
        ctx["kv_lens"], ctx["ropes"], ctx["past_key_values"] = kv_lens, ropes, pkv
        return ctx

    @torch.no_grad()
    def gen_text_batch(self, ctx, max_length: int = 500, do_sample: bool = True, temperature: float = 1.0) -> List[str]:
        """gen_text for every sample of a batched context; each answer ends at that sample's own EOS."""
        cache = ctx["past_key_values"]
        eos = self.new_token_ids["eos_token_id"]
        gi = self.model.prepare_start_tokens(ctx["kv_lens"], ctx["ropes"], self.new_token_ids)
        lens0 = list(cache.lens)
        try:        # decode in place beyond the committed lengths, then put the lengths back (see gen_text)
            ids = self.model.generate_text(past_key_values=cache, max_length=max_length, do_sample=do_sample,
                                           temperature=temperature, end_token_id=eos, per_sample_eos=True, **gi).cpu()
        finally:
            if cache.slabs is not None:
                cache.lens = lens0
        out = []
        for b in range(ids.shape[1]):
            col = ids[:, b]
            hit = (col[1:] == eos).nonzero()
            n = int(hit[0]) + 1 if hit.numel() else col.numel()   # rows fed before this sample's EOS (row 0 = start token)
            text = self.tokenizer.decode(col[:n])
            out.append(text.split("<|im_end|>")[0].split("<|im_start|>")[1])
        return out

    @torch.no_grad()
    def gen_image_batch(self, image_shapes, ctx, cfg_text_precontext, cfg_img_precontext, cfg_text_scale=4.0, cfg_img_scale=1.5,
                        cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global", num_timesteps=50,
                        timestep_shift=3.0) -> List[Image.Image]:
        m = self.model
        n = len(ctx["kv_lens"])
        shapes = [tuple(image_shapes)] * n if isinstance(image_shapes[0], int) else [tuple(s) for s in image_shapes]
        gi = m.prepare_vae_latent(curr_kvlens=ctx["kv_lens"], curr_rope=ctx["ropes"], image_sizes=shapes,
                                  new_token_ids=self.new_token_ids)
        gt = m.prepare_vae_latent_cfg(curr_kvlens=cfg_text_precontext["kv_lens"], curr_rope=cfg_text_precontext["ropes"],
                                      image_sizes=shapes)
        gim = m.prepare_vae_latent_cfg(curr_kvlens=cfg_img_precontext["kv_lens"], curr_rope=cfg_img_precontext["ropes"],
                                       image_sizes=shapes)
        latents = m.generate_image(
            past_key_values=ctx["past_key_values"], cfg_text_past_key_values=cfg_text_precontext["past_key_values"],
            cfg_img_past_key_values=cfg_img_precontext["past_key_values"], num_timesteps=num_timesteps,
            cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval,
            cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type, timestep_shift=timestep_shift, **gi,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
            cfg_img_packed_position_ids=gim["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gim["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=gim["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gim["cfg_packed_key_value_indexes"])
        return [self.decode_image(lat, shp) for lat, shp in zip(latents, shapes)]

    @torch.no_grad()
    def batch_interleave_inference(self, input_lists: List[List[Union[str, Image.Image]]], think=False,
                                   understanding_output=False, max_think_token_n=1000, do_sample=False, text_temperature=0.3,
                                   cfg_text_scale=3.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), timestep_shift=3.0,
                                   num_timesteps=50, cfg_renorm_min=0.0, cfg_renorm_type="global", image_shapes=(1024, 1024)
                                   ) -> List[List[Union[str, Image.Image]]]:
        """interleave_inference (inferencer.py:552-638) over B samples at once.  Every sample must have the same
        sequence of item types; lengths (prompt tokens, image sizes) may differ.  Returns one output list per sample."""
        n = len(input_lists)
        if n == 0:
            return []
        kinds = [tuple("s" if isinstance(t, str) else "i" if isinstance(t, Image.Image) else "?" for t in items) for items in input_lists]
        if any("?" in k for k in kinds):
            raise ValueError("Unsupported input type in a batched input list")
        if len(set(kinds)) != 1:
            raise ValueError("batched samples must share one item structure (e.g. every sample = [image, text])")
        need_cfg = not understanding_output
        ctx = self._batch_context(n)
        cfg_text_ctx, cfg_img_ctx = self._snap(ctx), self._snap(ctx)     # prefix sharing as in interleave_inference
        img_in_sync = True
        if think:
            sp = VLM_THINK_SYSTEM_PROMPT if understanding_output else GEN_THINK_SYSTEM_PROMPT
            ctx = self._update_batch_text([sp] * n, ctx)
            if need_cfg:
                cfg_img_ctx = self._snap(ctx)
        for j, kind in enumerate(kinds[0]):
            column = [items[j] for items in input_lists]
            if kind == "s":
                if need_cfg:
                    cfg_text_ctx = self._snap(ctx)
                ctx = self._update_batch_text(column, ctx)
                if need_cfg:
                    cfg_img_ctx = self._snap(ctx) if img_in_sync else self._update_batch_text(column, cfg_img_ctx)
            else:
                column = [self.vae_transform.resize_transform(pil_img2rgb(im)) for im in column]
                ctx = self._update_batch_image(column, ctx, vae=not understanding_output)
                img_in_sync = False
                if need_cfg:
                    cfg_text_ctx = self._snap(ctx)
        outputs = [[] for _ in range(n)]
        if understanding_output:
            for o, t in zip(outputs, self.gen_text_batch(ctx, do_sample=do_sample, temperature=text_temperature,
                                                         max_length=max_think_token_n)):
                o.append(t)
            return outputs
        if think:
            thoughts = self.gen_text_batch(ctx, do_sample=do_sample, temperature=text_temperature, max_length=max_think_token_n)
            ctx = self._update_batch_text(thoughts, ctx)
            for o, t in zip(outputs, thoughts):
                o.append(t)
        images = self.gen_image_batch(image_shapes, ctx, cfg_text_precontext=cfg_text_ctx, cfg_img_precontext=cfg_img_ctx,
                                      cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval,
                                      timestep_shift=timestep_shift, num_timesteps=num_timesteps, cfg_renorm_min=cfg_renorm_min,
                                      cfg_renorm_type=cfg_renorm_type)
        for o, im in zip(outputs, images):
            o.append(im)
        return outputs

    @staticmethod
    def _collect(output_list):
        output_dict = {"image": None, "text": None}
        for item in output_list:
            if isinstance(item, Image.Image):
                if output_dict["image"] is None:
                    output_dict["image"] = []
                output_dict["image"].append(item)
            elif isinstance(item, str):
                output_dict["text"] = item
        if isinstance(output_dict["image"], list) and len(output_dict["image"]) == 1:
            output_dict["image"] = output_dict["image"][0]
        return output_dict

    def __call__(self, image: Optional[Union[Image.Image, List[Image.Image]]] = None, text: Optional[str] = None,
                 inference_ver=0, **kargs) -> Dict[str, Any]:
        """inferencer.py:640-680: images first, then the text; {'image': PIL|list|None, 'text': str|None}.
        Batch extension: `text` as a LIST of prompts (and `image` as a list of the same length whose entries are a PIL
        image, a list of PIL images, or None for all) runs the samples together and returns a list of such dicts."""
        if isinstance(text, (list, tuple)):
            if inference_ver != 0:
                raise ValueError("the batched call supports inference_ver=0 only")
            n = len(text)
            if image is None:
                image = [None] * n
            if not isinstance(image, (list, tuple)) or len(image) != n:
                raise ValueError("batched call: `image` must be None or a list with one entry per prompt")
            input_lists = []
            for im, tx in zip(image, text):
                items = [] if im is None else (list(im) if isinstance(im, (list, tuple)) else [im])
                input_lists.append(items + [tx])
            return [self._collect(o) for o in self.batch_interleave_inference(input_lists, **kargs)]
        output_dict = {"image": None, "text": None}
        if image is None and text is None:
            return output_dict
        input_list = []
        if image is not None:
            input_list.extend(image if isinstance(image, list) else [image])
        if text is not None:
            input_list.append(text)
        if inference_ver == 0:
            output_list = self.interleave_inference(input_list, **kargs)
        elif inference_ver == 1:
            output_list = self.interleave_inference_for_vqa_reconstruction_ver1(input_list, **kargs)
        else:
            raise ValueError(f"Unsupported inference_ver: {inference_ver}")
        for item in output_list:
            if isinstance(item, Image.Image):
                if output_dict["image"] is None:
                    output_dict["image"] = []
                output_dict["image"].append(item)
            elif isinstance(item, str):
                output_dict["text"] = item
        if isinstance(output_dict["image"], list) and len(output_dict["image"]) == 1:
            output_dict["image"] = output_dict["image"][0]
        return output_dict
