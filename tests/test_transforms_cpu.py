"""unimedvl_amd.transforms against fixtures made by the REFERENCE's data/transforms.py (oracle/gen_golden.py section I: the
reference module imported over torchvision / cv2 stand-ins, SURVEY.md section 8c shim 3): the size arithmetic of
MaxLongEdgeMinShortEdgeResize.forward (transforms.py:58-86) over 6 parameter sets x 17 image sizes x 3 img_num values,
and the full ImageTransform (resize -> ToTensor -> Normalize, transforms.py:90-115) bit for bit on RGB / L / RGBA inputs,
a 600x437 image at the VQA script's parameters, and tensor inputs."""
import hashlib

import numpy as np
import torch
from PIL import Image

from conftest import load_golden


def test_resize_target_sizes_match_reference():
    from unimedvl_amd.transforms import MaxLongEdgeMinShortEdgeResize
    g = load_golden("transforms")
    psets = g["psets"].tolist()
    rz = [MaxLongEdgeMinShortEdgeResize(*p) for p in psets]
    n = 0
    for (pi, w, h, img_num), (ow, oh) in zip(g["size_in"].tolist(), g["size_out"].tolist()):
        assert rz[pi].target_size(w, h, img_num=img_num) == (ow, oh), (psets[pi], w, h, img_num)
        assert ow % psets[pi][2] == 0 and oh % psets[pi][2] == 0
        n += 1
    assert n == len(psets) * 17 * 3
    out = rz[0](Image.new("L", (1500, 2000)), img_num=2)          # the PIL path returns an image of exactly that size
    i = g["size_in"].tolist().index([0, 1500, 2000, 2])
    assert list(out.size) == g["size_out"].tolist()[i]


def test_image_transform_bit_exact():
    from unimedvl_amd.data_utils import pil_img2rgb
    from unimedvl_amd.transforms import ImageTransform
    g = load_golden("transforms")
    t_small, t_vae = ImageTransform(56, 28, 14), ImageTransform(64, 32, 16)
    rgb = Image.fromarray(g["rgb"].numpy())
    gray = Image.fromarray(g["gray"].numpy())
    rgba = Image.fromarray(g["rgba"].numpy(), mode="RGBA")
    assert torch.equal(t_small(pil_img2rgb(rgb)), g["rgb_vit"])
    assert torch.equal(t_small(pil_img2rgb(gray)), g["gray_vit"])
    assert torch.equal(t_vae(pil_img2rgb(rgba)), g["rgba_vae"])        # alpha composited on white (data_utils.py:116-137)
    assert torch.equal(t_small(pil_img2rgb(rgb), img_num=3), g["rgb_vit_num3"])
    big = Image.fromarray(np.random.default_rng(78).integers(0, 256, (600, 437, 3), dtype=np.uint8))
    out = ImageTransform(980, 378, 14, max_pixels=2_007_040)(big)      # interactive_vqa_inferencer.py's parameters
    assert list(out.shape) == g["big_shape"].tolist()
    assert hashlib.sha256(out.contiguous().numpy().tobytes()).hexdigest() == g["big_sha256"]
    assert float(out.min()) >= -1.0 and float(out.max()) <= 1.0


def test_resize_accepts_tensors_like_the_reference():
    from unimedvl_amd.transforms import ImageTransform
    g = load_golden("transforms")
    t_small, t_vae = ImageTransform(56, 28, 14), ImageTransform(64, 32, 16)
    got = t_small.resize_transform(g["u8_in"])
    assert got.dtype == torch.uint8 and torch.equal(got, g["u8_resized"])
    got = t_vae.resize_transform(g["f32_in"])
    assert got.dtype == torch.float32 and torch.equal(got, g["f32_resized"])
    assert t_vae.resize_transform.max_size == 64 and t_vae.resize_transform.stride == 16      # read by inferencer.py:44-47
