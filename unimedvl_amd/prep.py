"""Host-side packing: the reference's ``Bagel.prepare_*`` functions
(codes/modeling/unimedvl/bagel.py:377-409, 460-520, 617-694, 809-898, 1213-1233).

They turn prompts / images / target sizes into the packed index tensors that ARE the
boundary's data format (``generation_input`` dicts, same keys, dtypes and values as the
reference) and advance the (kv_lens, ropes) counters: an image span shares ONE rope
position, text advances by its length.  Pure host code - usable without a GPU.
"""
import torch

from .config import UniMedVLConfig
from .data_utils import (PackedVitImages, get_flattened_position_ids_extrapolate, get_flattened_position_ids_interpolate, patchify)


class BagelPrep:
    def __init__(self, cfg: UniMedVLConfig, interpolate_pos=False):
        self.cfg = cfg
        self.latent_patch_size = cfg.latent_patch
        self.latent_downsample = cfg.latent_downsample
        self.max_latent_size = cfg.max_latent
        self.latent_channel = cfg.z_channels
        self.vit_patch_size = cfg.patch
        self.vit_max_num_patch_per_side = cfg.vit_side
        self.get_flattened_position_ids = (get_flattened_position_ids_interpolate if interpolate_pos
                                           else get_flattened_position_ids_extrapolate)

    def prepare_prompts(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        text_ids, pos_ids, lens, text_idx, kv_idx = [], [], [], [], []
        curr = 0
        newlens, new_rope = [], []
        for prompt, kvlen, rope in zip(prompts, curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen))
            curr += kvlen
            ids = [new_token_ids["bos_token_id"]] + list(tokenizer.encode(prompt)) + [new_token_ids["eos_token_id"]]
            lens.append(len(ids))
            text_ids.extend(ids)
            pos_ids.extend(range(rope, rope + len(ids)))
            text_idx.extend(range(curr, curr + len(ids)))
            newlens.append(kvlen + len(ids))
            new_rope.append(rope + len(ids))
            curr += len(ids)
        generation_input = {
            "text_token_lens": torch.tensor(lens, dtype=torch.int),
            "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
            "packed_text_position_ids": torch.tensor(pos_ids, dtype=torch.long),
            "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
            "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
        }
        return generation_input, newlens, new_rope

    def prepare_vit_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids):
        vit_idx, vit_lens, vit_tokens, vit_pos = [], [], [], []
        text_ids, text_idx = [], []
        seqlens, pos_ids, indexes, kv_idx = [], [], [], []
        _curr = curr = 0
        newlens, new_rope = [], []
        for image, kvlen, rope in zip(images, curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen))
            curr += kvlen
            text_ids.append(new_token_ids["start_of_image"])
            text_idx.append(_curr)
            indexes.append(curr)
            curr += 1
            _curr += 1
            image_tensor = transforms(image)
            vit_pos.append(self.get_flattened_position_ids(image_tensor.size(1), image_tensor.size(2), self.vit_patch_size,
                                                           max_num_patches_per_side=self.vit_max_num_patch_per_side))
            if getattr(self, "device_patchify", False):      # the engine patchifies on the device (PackedVitImages below)
                vit_tokens.append(image_tensor)
                n = (image_tensor.size(1) // self.vit_patch_size) * (image_tensor.size(2) // self.vit_patch_size)
            else:
                toks = patchify(image_tensor, self.vit_patch_size)
                vit_tokens.append(toks)
                n = toks.shape[0]
            vit_lens.append(n)
            vit_idx.extend(range(_curr, _curr + n))
            indexes.extend(range(curr, curr + n))
            curr += n
            _curr += n
            text_ids.append(new_token_ids["end_of_image"])
            text_idx.append(_curr)
            indexes.append(curr)
            curr += 1
            _curr += 1
            pos_ids.extend([rope] * (n + 2))
            seqlens.append(n + 2)
            newlens.append(kvlen + n + 2)
            new_rope.append(rope + 1)
        generation_input = {
            "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
            "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
            "vit_token_seqlens": torch.tensor(vit_lens, dtype=torch.int),
            # the reference's [sum tokens, 3 p^2] tensor, or - device_patchify - a stand-in that IS that tensor for whoever asks and
            # hands the engine the images
            "packed_vit_tokens": (PackedVitImages(vit_tokens, self.vit_patch_size) if getattr(self, "device_patchify", False)
                                  else torch.cat(vit_tokens, dim=0)),
            "packed_vit_position_ids": torch.cat(vit_pos, dim=0),
            "packed_vit_token_indexes": torch.tensor(vit_idx, dtype=torch.long),
            "packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
            "packed_seqlens": torch.tensor(seqlens, dtype=torch.int),
            "packed_indexes": torch.tensor(indexes, dtype=torch.long),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
            "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
        }
        return generation_input, newlens, new_rope

    def prepare_vae_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids, timestep=0):
        shapes, vae_pos, vae_idx = [], [], []
        text_ids, text_idx = [], []
        seqlens, pos_ids, indexes, kv_idx = [], [], [], []
        _curr = curr = 0
        tensors = []
        newlens, new_rope = [], []
        for image, kvlen, rope in zip(images, curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen))
            curr += kvlen
            text_ids.append(new_token_ids["start_of_image"])
            text_idx.append(_curr)
            indexes.append(curr)
            curr += 1
            _curr += 1
            image_tensor = transforms(image)
            tensors.append(image_tensor)
            vae_pos.append(self.get_flattened_position_ids(image_tensor.size(1), image_tensor.size(2),
                                                           self.latent_downsample,
                                                           max_num_patches_per_side=self.max_latent_size))
            H, W = image_tensor.shape[1:]
            h, w = H // self.latent_downsample, W // self.latent_downsample
            shapes.append((h, w))
            n = h * w
            vae_idx.extend(range(_curr, _curr + n))
            indexes.extend(range(curr, curr + n))
            curr += n
            _curr += n
            text_ids.append(new_token_ids["end_of_image"])
            text_idx.append(_curr)
            indexes.append(curr)
            curr += 1
            _curr += 1
            pos_ids.extend([rope] * (n + 2))
            seqlens.append(n + 2)
            newlens.append(kvlen + n + 2)
            new_rope.append(rope + 1)
        sizes = [t.shape for t in tensors]
        max_size = [max(s) for s in zip(*sizes)]
        padded = torch.zeros(size=(len(tensors), *max_size))
        for i, t in enumerate(tensors):
            padded[i, :, :t.shape[1], :t.shape[2]] = t
        generation_input = {
            "padded_images": padded,
            "patchified_vae_latent_shapes": shapes,
            "packed_vae_position_ids": torch.cat(vae_pos, dim=0),
            "packed_timesteps": torch.tensor([timestep]),
            "packed_vae_token_indexes": torch.tensor(vae_idx, dtype=torch.long),
            "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
            "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
            "packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
            "packed_seqlens": torch.tensor(seqlens, dtype=torch.int),
            "packed_indexes": torch.tensor(indexes, dtype=torch.long),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
            "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
        }
        return generation_input, newlens, new_rope

    def prepare_vae_latent(self, curr_kvlens, curr_rope, image_sizes, new_token_ids):
        text_ids, text_idx = [], []
        vae_pos, vae_idx, noises = [], [], []
        pos_ids, seqlens, indexes, kv_idx = [], [], [], []
        query_curr = curr = 0
        for (H, W), kvlen, rope in zip(image_sizes, curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen))
            curr += kvlen
            text_ids.append(new_token_ids["start_of_image"])
            text_idx.append(query_curr)
            indexes.append(curr)
            curr += 1
            query_curr += 1
            vae_pos.append(self.get_flattened_position_ids(H, W, self.latent_downsample,
                                                           max_num_patches_per_side=self.max_latent_size))
            h, w = H // self.latent_downsample, W // self.latent_downsample
            n = h * w
            noises.append(torch.randn(n, self.latent_channel * self.latent_patch_size ** 2))   # CPU RNG, as the reference
            vae_idx.extend(range(query_curr, query_curr + n))
            indexes.extend(range(curr, curr + n))
            curr += n
            query_curr += n
            text_ids.append(new_token_ids["end_of_image"])
            text_idx.append(query_curr)
            indexes.append(curr)
            curr += 1
            query_curr += 1
            pos_ids.extend([rope] * (n + 2))
            seqlens.append(n + 2)
        return {
            "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
            "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
            "packed_init_noises": torch.cat(noises, dim=0),
            "packed_vae_position_ids": torch.cat(vae_pos, dim=0),
            "packed_vae_token_indexes": torch.tensor(vae_idx, dtype=torch.long),
            "packed_seqlens": torch.tensor(seqlens, dtype=torch.int),
            "packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
            "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
            "packed_indexes": torch.tensor(indexes, dtype=torch.long),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
        }

    def prepare_vae_latent_cfg(self, curr_kvlens, curr_rope, image_sizes):
        pos_ids, indexes, kv_idx = [], [], []
        curr = 0
        for (H, W), kvlen, rope in zip(image_sizes, curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen))
            curr += kvlen
            n = (H // self.latent_downsample) * (W // self.latent_downsample)
            indexes.extend(range(curr, curr + n + 2))
            curr += n + 2
            pos_ids.extend([rope] * (n + 2))
        return {
            "cfg_packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
            "cfg_key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
            "cfg_packed_query_indexes": torch.tensor(indexes, dtype=torch.long),
            "cfg_packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
        }

    def prepare_start_tokens(self, curr_kvlens, curr_rope, new_token_ids):
        start, kv_idx, pos = [], [], []
        curr = 0
        for kvlen, rope in zip(curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen))
            start.append(new_token_ids["bos_token_id"])
            pos.append(rope)
            curr += kvlen
        return {
            "packed_start_tokens": torch.tensor(start, dtype=torch.long),
            "packed_query_position_ids": torch.tensor(pos, dtype=torch.long),
            "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
        }
