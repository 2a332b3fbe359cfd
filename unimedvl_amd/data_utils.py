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


class PackedVitImages:
    """What `prepare_vit_images` puts under "packed_vit_tokens" when the engine patchifies on the device: the transformed
    [3, H, W] images themselves.  The reference's tensor - torch.cat([patchify(im, p) ...]), data_utils.py:43-50 /
    bagel.py:540-548 - is what `tokens()` returns, bit for bit; torch functions, attribute access and indexing on this object
    all go through it, so code written against the reference's `generation_input` keeps working.  The engine never asks for
    it: it uploads the images and runs umv_patchify_f32_bf16 (the host-side permute costs 4 ms per 448 x 448 image)."""

    def __init__(self, images, patch_size):
        self.images = [im.contiguous() for im in images]
        self.patch_size = int(patch_size)
        self._tokens = None

    def tokens(self):
        """The reference's packed patch tensor, made where the images live and in their dtype - so `.to("cuda")` followed by a
        torch function behaves like the move of a tensor would (same device, same dtype as the reference's result)."""
        if self._tokens is None:
            self._tokens = torch.cat([patchify(im, self.patch_size) for im in self.images], dim=0)
        return self._tokens

    @property
    def device(self):
        return self.images[0].device if self.images else torch.device("cpu")

    @property
    def dtype(self):
        return self.images[0].dtype if self.images else torch.float32

    @property
    def is_cuda(self):
        return self.device.type == "cuda"

    def token_counts(self):
        p = self.patch_size
        return [(im.shape[1] // p) * (im.shape[2] // p) for im in self.images]

    @property
    def shape(self):
        return torch.Size((sum(self.token_counts()), self.patch_size ** 2 * (self.images[0].shape[0] if self.images else 3)))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def to(self, *args, **kwargs):
        """device moves keep the images (dtype changes apply to the images as well)"""
        out = PackedVitImages([im.to(*args, **kwargs) for im in self.images], self.patch_size)
        return out

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return self.tokens()[idx]

    def __getattr__(self, name):          # anything else a tensor has: the reference's tensor answers
        if name.startswith("_") or name in ("images", "patch_size"):
            raise AttributeError(name)
        return getattr(self.tokens(), name)

    def __array__(self, *a, **k):
        return self.tokens().numpy().__array__(*a, **k)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(v):
            if isinstance(v, PackedVitImages):
                return v.tokens()
            if isinstance(v, (list, tuple)):
                return type(v)(unwrap(u) for u in v)
            return v
        return func(*unwrap(tuple(args)), **{k: unwrap(v) for k, v in (kwargs or {}).items()})


def _delegate(name):
    def op(self, *a, **k):
        return getattr(self.tokens(), name)(*a, **k)
    op.__name__ = name
    return op


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__neg__", "__matmul__",
           "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__iter__", "__float__", "__int__", "__bool__"):
    setattr(PackedVitImages, _n, _delegate(_n))
PackedVitImages.__hash__ = object.__hash__


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
