"""Inference-time image transform with the reference's interface
(codes/data/transforms.py:15-115): ``ImageTransform(max, min, stride)(pil) -> [3,H,W]`` in
[-1,1] and ``.resize_transform`` exposing max_size / min_size / stride / max_pixels
(read by InterleaveInferencer._calculate_target_size_with_aspect_ratio).  No torchvision:
its F.resize on a PIL image is PIL's own BICUBIC resize, on a tensor torch's antialiased
bicubic interpolate.  Pinned to the reference's own module by tests/golden/transforms.npz
(tests/test_transforms_cpu.py)."""
import numpy as np
import torch
from PIL import Image


class MaxLongEdgeMinShortEdgeResize:
    def __init__(self, max_size, min_size, stride, max_pixels, interpolation=Image.BICUBIC, antialias=True):
        self.max_size, self.min_size, self.stride, self.max_pixels = max_size, min_size, stride, max_pixels
        self.interpolation, self.antialias = interpolation, antialias

    def _make_divisible(self, value, stride):
        return max(stride, int(round(value / stride) * stride))

    def _apply_scale(self, width, height, scale):
        return (self._make_divisible(round(width * scale), self.stride),
                self._make_divisible(round(height * scale), self.stride))

    def target_size(self, width, height, img_num=1):
        scale = min(self.max_size / max(width, height), 1.0)
        scale = max(scale, self.min_size / min(width, height))
        nw, nh = self._apply_scale(width, height, scale)
        if nw * nh > self.max_pixels / img_num:
            nw, nh = self._apply_scale(nw, nh, self.max_pixels / img_num / (nw * nh))
        if max(nw, nh) > self.max_size:
            nw, nh = self._apply_scale(nw, nh, self.max_size / max(nw, nh))
        return nw, nh

    def __call__(self, img, img_num=1):
        if isinstance(img, torch.Tensor):
            # tensor input [..., H, W] (transforms.py:64-65 -> torchvision's tensor resize): bicubic, align_corners=False,
            # antialias; uint8 inputs are rounded and clamped back
            height, width = img.shape[-2:]
            nw, nh = self.target_size(width, height, img_num=img_num)
            x = img if img.dim() == 4 else img.unsqueeze(0)
            out = torch.nn.functional.interpolate(x.to(torch.float32), size=(nh, nw), mode="bicubic", align_corners=False,
                                                  antialias=bool(self.antialias))
            out = out.round().clamp(0, 255).to(torch.uint8) if img.dtype == torch.uint8 else out.to(img.dtype)
            return out if img.dim() == 4 else out[0]
        nw, nh = self.target_size(*img.size, img_num=img_num)
        return img.resize((nw, nh), self.interpolation)

    forward = __call__


class ImageTransform:
    def __init__(self, max_image_size, min_image_size, image_stride, max_pixels=14 * 14 * 9 * 1024,
                 image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5)):
        self.stride = image_stride
        self.resize_transform = MaxLongEdgeMinShortEdgeResize(max_image_size, min_image_size, image_stride, max_pixels)
        self.mean = torch.tensor(image_mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(image_std, dtype=torch.float32).view(3, 1, 1)

    def __call__(self, img, img_num=1):
        img = self.resize_transform(img, img_num=img_num)
        arr = np.asarray(img.convert("RGB"), dtype=np.uint8)
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)   # ToTensor
        return t.sub_(self.mean).div_(self.std)                                       # Normalize(0.5, 0.5)
