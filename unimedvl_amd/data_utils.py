"""Host-side preprocessing helpers with the reference's names and semantics
(codes/data/data_utils.py:43-58,116-175)."""
import torch
from PIL import Image


def patchify(image, patch_size):
    """[C,H,W] -> [(H/p)*(W/p), p*p*C] in (h, w, p, q, c) order (data_utils.py:43-50)."""
    p = patch_size
    c, h, w = image.shape
    assert h % p == 0 and w % p == 0
    image = image.reshape(c, h // p, p, w // p, p)
    return image.permute(1, 3, 2, 4, 0).reshape(-1, p * p * c)


def get_flattened_position_ids_extrapolate(img_h, img_w, patch_size, max_num_patches_per_side):
    """row * max_side + col (data_utils.py:53-58)."""
    nh, nw = img_h // patch_size, img_w // patch_size
    return (torch.arange(nh)[:, None] * max_num_patches_per_side + torch.arange(nw)).flatten()


def get_flattened_position_ids_interpolate(img_h, img_w, patch_size, max_num_patches_per_side):
    """bucketised fractional coordinates (data_utils.py:61-69)."""
    nh, nw = img_h // patch_size, img_w // patch_size
    bounds = torch.arange(1 / max_num_patches_per_side, 1.0, 1 / max_num_patches_per_side)
    fh = torch.arange(0, 1 - 1e-6, 1 / nh)
    fw = torch.arange(0, 1 - 1e-6, 1 / nw)
    bh = torch.bucketize(fh, bounds, right=True)
    bw = torch.bucketize(fw, bounds, right=True)
    return (bh[:, None] * max_num_patches_per_side + bw).flatten()


def pil_img2rgb(image):
    """RGBA / palette-with-transparency composited on white, everything else .convert('RGB')
    (data_utils.py:116-137)."""
    width, height = image.size
    if width * height > 20_000_000:
        raise ValueError(f"Image too large: {width * height} pixels")
    if image.mode == "RGBA" or image.info.get("transparency", None) is not None:
        image = image.convert("RGBA")
        white = Image.new(mode="RGB", size=image.size, color=(255, 255, 255))
        white.paste(image, mask=image.split()[3])
        return white
    return image.convert("RGB")


def add_special_tokens(tokenizer):
    """Registers <|im_start|>, <|im_end|>, <|vision_start|>, <|vision_end|> if missing and
    returns (tokenizer, new_token_ids, num_new_tokens) (data_utils.py:140-175)."""
    present = []
    for v in tokenizer.special_tokens_map.values():
        present += [v] if isinstance(v, str) else list(v)
    wanted = ["<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>"]
    num_new = tokenizer.add_tokens([t for t in wanted if t not in present])
    ids = [tokenizer.convert_tokens_to_ids(t) for t in wanted]
    new_token_ids = dict(bos_token_id=ids[0], eos_token_id=ids[1], start_of_image=ids[2], end_of_image=ids[3])
    return tokenizer, new_token_ids, num_new
