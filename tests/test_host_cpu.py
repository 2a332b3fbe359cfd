"""CPU-side checks (no GPU): host packing against the reference's own prepare_* outputs
(bit-exact integer/index work), preprocessing helpers, the C-ABI surface, and the loud
failure of the product path without the HIP extension / a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_golden, NEW_TOKEN_IDS, ROOT


class ListTokenizer:
    def encode(self, s):
        return [int(x) for x in s.split()]


@pytest.fixture(scope="module")
def prep(tiny_weights):
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.prep import BagelPrep
    return BagelPrep(UniMedVLConfig.from_dict(tiny_weights[0]))


def same(gi, g, prefix):
    keys = [k[len(prefix):] for k in g if k.startswith(prefix)]
    assert keys, prefix
    for k in keys:
        ref = g[prefix + k]
        got = gi[k]
        if isinstance(got, list):
            got = torch.tensor(got)
        assert got.dtype == ref.dtype or ref.dtype == torch.int64, (k, got.dtype, ref.dtype)
        assert torch.equal(got.to(ref.dtype), ref), f"{prefix}{k} differs from the reference"
    tensor_keys = {k for k, v in gi.items() if torch.is_tensor(v) or isinstance(v, list)}
    assert tensor_keys == set(keys), (tensor_keys ^ set(keys))


def test_prepare_functions_match_reference(prep):
    g = load_golden("prep")
    ident = lambda x: x
    tok = ListTokenizer()
    c = g["counters"].tolist()
    gi, kv1, rp1 = prep.prepare_vit_images([3, 0], [2, 0], [g["image0"], g["image1"]], ident, NEW_TOKEN_IDS)
    same(gi, g, "vit.")
    assert [kv1, rp1] == c[0:2]
    gi, kv2, rp2 = prep.prepare_prompts(kv1, rp1, ["5 6 7 8 9 10 11", "200 100"], tok, NEW_TOKEN_IDS)
    same(gi, g, "txt.")
    assert [kv2, rp2] == c[2:4]
    gi, kv3, rp3 = prep.prepare_vae_images(kv2, rp2, [g["vimage0"], g["vimage1"]], ident, NEW_TOKEN_IDS)
    same(gi, g, "vae.")
    assert [kv3, rp3] == c[4:6]
    torch.manual_seed(77)   # the reference draws the init noise from the global CPU RNG (bagel.py:835-837)
    gi = prep.prepare_vae_latent(kv3, rp3, [(64, 64), (32, 48)], NEW_TOKEN_IDS)
    same(gi, g, "lat.")
    same(prep.prepare_vae_latent_cfg(kv1, rp1, [(64, 64), (32, 48)]), g, "cfg.")
    same(prep.prepare_start_tokens(kv3, rp3, NEW_TOKEN_IDS), g, "start.")


def test_prepare_edge_cases(prep):
    tok = ListTokenizer()
    gi, kv, rp = prep.prepare_prompts([0], [0], [""], tok, NEW_TOKEN_IDS)          # empty prompt -> bos, eos only
    assert gi["packed_text_ids"].tolist() == [NEW_TOKEN_IDS["bos_token_id"], NEW_TOKEN_IDS["eos_token_id"]]
    assert kv == [2] and rp == [2]
    gi, kv, rp = prep.prepare_prompts([], [], [], tok, NEW_TOKEN_IDS)              # empty batch
    assert gi["packed_text_ids"].numel() == 0 and kv == [] and rp == []
    img = torch.zeros(3, 14, 14)                                                    # one-patch image
    gi, kv, rp = prep.prepare_vit_images([5], [4], [img], lambda x: x, NEW_TOKEN_IDS)
    assert gi["packed_seqlens"].tolist() == [3] and kv == [8] and rp == [5]
    assert gi["packed_position_ids"].tolist() == [4, 4, 4]                          # an image span shares one rope position
    with pytest.raises(AssertionError):
        prep.prepare_vit_images([0], [0], [torch.zeros(3, 15, 14)], lambda x: x, NEW_TOKEN_IDS)


def test_patchify_and_position_ids():
    from unimedvl_amd.data_utils import patchify, get_flattened_position_ids_extrapolate, get_flattened_position_ids_interpolate
    img = torch.arange(3 * 28 * 42, dtype=torch.float32).reshape(3, 28, 42)
    p = patchify(img, 14)
    ref = torch.einsum("chpwq->hwpqc", img.reshape(3, 2, 14, 3, 14)).reshape(-1, 588)   # data_utils.py:47-49
    assert torch.equal(p, ref)
    assert get_flattened_position_ids_extrapolate(28, 42, 14, 70).tolist() == [0, 1, 2, 70, 71, 72]
    ids = get_flattened_position_ids_interpolate(28, 28, 14, 4)
    assert ids.tolist() == [0, 2, 8, 10]


def test_transforms_and_rgb():
    from PIL import Image
    from unimedvl_amd.data_utils import pil_img2rgb
    from unimedvl_amd.transforms import ImageTransform
    tf = ImageTransform(980, 378, 14, max_pixels=2_007_040)          # data/default.yaml vlm_sft numbers
    assert tf.resize_transform.target_size(448, 448) == (448, 448)
    assert tf.resize_transform.target_size(2000, 1000) == (980, 490)
    w, h = tf.resize_transform.target_size(100, 300)
    assert w % 14 == 0 and h % 14 == 0 and max(w, h) <= 980
    rgba = Image.new("RGBA", (20, 10), (255, 0, 0, 0))
    out = pil_img2rgb(rgba)
    assert out.mode == "RGB" and out.getpixel((0, 0)) == (255, 255, 255)   # transparent -> white
    t = ImageTransform(64, 32, 16)(Image.new("L", (40, 50), 255))
    assert t.shape[0] == 3 and t.shape[1] % 16 == 0 and t.shape[2] % 16 == 0 and float(t.max()) == 1.0


def test_packed_vit_images_is_the_reference_tensor_for_whoever_asks():
    """data_utils.PackedVitImages (what prepare_vit_images returns as "packed_vit_tokens" when the engine patchifies on the device)
    must behave as the reference's tensor: same values through torch functions, indexing, attributes; .to() keeps the images."""
    from unimedvl_amd.data_utils import PackedVitImages, patchify
    from unimedvl_amd.prep import BagelPrep
    g = torch.Generator().manual_seed(5)
    imgs = [torch.randn(3, 28, 42, generator=g), torch.randn(3, 56, 14, generator=g)]
    ref = torch.cat([patchify(im, 14) for im in imgs], 0)
    pv = PackedVitImages(imgs, 14)
    assert tuple(pv.shape) == tuple(ref.shape) and len(pv) == ref.shape[0] and pv.size(1) == 588 and pv.token_counts() == [6, 4]
    assert torch.equal(pv, ref) and torch.equal(pv[3:7], ref[3:7]) and torch.equal(torch.cat([pv, pv], 0), torch.cat([ref, ref], 0))
    assert pv.dtype == torch.float32 and float(pv.abs().sum()) == float(ref.abs().sum())
    assert torch.equal((pv * 2.0), ref * 2.0) and torch.equal(pv.numpy().sum() + torch.zeros(()), ref.numpy().sum() + torch.zeros(()))
    moved = pv.to("cpu")
    assert isinstance(moved, PackedVitImages) and torch.equal(moved.tokens(), ref)
    # a dtype move applies to the tokens like it would to the reference's tensor; .device follows the images
    half = pv.to(torch.bfloat16)
    assert half.dtype == torch.bfloat16 and half.tokens().dtype == torch.bfloat16 and torch.equal(half.tokens(), ref.to(torch.bfloat16))
    assert pv.device == ref.device and not pv.is_cuda

    class P(BagelPrep):                       # the prep mix-in alone (no device): both contracts from the same call
        vit_patch_size, vit_max_num_patch_per_side, device_patchify = 14, 70, False

        def get_flattened_position_ids(self, h, w, p, max_num_patches_per_side):
            from unimedvl_amd.data_utils import get_flattened_position_ids_extrapolate as f
            return f(h, w, p, max_num_patches_per_side)
    prep = P.__new__(P)
    ntid = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)
    a, kv_a, rope_a = prep.prepare_vit_images([0, 5], [0, 5], imgs, lambda x: x, ntid)
    prep.device_patchify = True
    b, kv_b, rope_b = prep.prepare_vit_images([0, 5], [0, 5], imgs, lambda x: x, ntid)
    assert kv_a == kv_b and rope_a == rope_b and isinstance(b["packed_vit_tokens"], PackedVitImages)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_c_abi_exports_every_declared_symbol():
    """Every function include/unimedvl_hip.h declares must be exported by the built library and
    bound by the ctypes layer (no compute call: this runs without a GPU)."""
    hdr = open(os.path.join(ROOT, "include", "unimedvl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(umv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from unimedvl_amd import _lib
    lib_path = _lib.LIB_PATH
    assert os.path.exists(lib_path), "build the library first: python -m unimedvl_amd.build"
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.declared_symbols()), declared ^ set(_lib.declared_symbols())
    assert _lib.load().umv_version() >= 100
    # the experimental package (experimental/): measured-and-not-adopted kernels, a library of its own that the product never loads
    from experimental import _lib as xlib
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "experimental", "include", "unimedvl_hip_experimental.h")).read(), flags=re.S)
    exp_declared = set(re.findall(r"\b(umv_[a-z0-9_]+)\s*\(", hdr))
    assert exp_declared and not (exp_declared & declared), exp_declared & declared
    assert exp_declared == set(xlib.declared_symbols())
    for name in sorted(exp_declared):
        assert not hasattr(lib, name), f"{name} is experimental but exported by the product library"
    if xlib.available():      # built only on request (python -m experimental.build)
        exp = ctypes.CDLL(xlib.EXP_LIB_PATH)
        for name in sorted(exp_declared):
            assert hasattr(exp, name), f"{name} declared in the experimental header but not exported"
    # nothing in the product package may import the experimental one
    for fn in os.listdir(os.path.join(ROOT, "unimedvl_amd")):
        if fn.endswith(".py"):
            assert "experimental" not in re.sub(r"#.*", "", open(os.path.join(ROOT, "unimedvl_amd", fn)).read()).replace("experimental/", ""), fn


def test_argument_errors_come_back_through_the_abi():
    from unimedvl_amd import _lib
    lib = _lib.load()
    rc = lib.umv_rmsnorm_bf16(None, None, None, None, None, 4, 64, 1e-6, None)
    assert rc < 0 and b"null pointer" in lib.umv_last_error()
    a = _lib.GemmArgs(x=1, ldx=8, wp=1, out=1, ldo=8, M=1, N=8, K=7, epilogue=0)
    assert lib.umv_gemm_bf16(ctypes.byref(a), None) < 0 and b"multiples of 8" in lib.umv_last_error()


def test_product_path_fails_loudly_without_gpu(tiny_weights):
    if torch.cuda.is_available():
        pytest.skip("has a GPU")
    from unimedvl_amd import ops
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    with pytest.raises(RuntimeError):
        Bagel(UniMedVLConfig.from_dict(tiny_weights[0]), lambda n: tiny_weights[1][n])
    with pytest.raises(_lib_error()):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-6)


def _lib_error():
    from unimedvl_amd._lib import UmvError
    return UmvError


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "unimedvl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_checkpoint_overlay_and_bf16_preference(tmp_path):
    """checkpoint.py: ema_bf16.safetensors is preferred over ema.safetensors (interactive_vqa_inferencer.py:127-147) and a
    fine-tuned checkpoint overlays the base one tensor by tensor (eval/vlm/utils.py:71-98); shape mismatches and missing
    tensors fail loudly."""
    import pytest
    import torch
    from safetensors.torch import save_file
    from unimedvl_amd.checkpoint import SafetensorsGetter, checkpoint_getter, find_weights_file
    base, ft = tmp_path / "base", tmp_path / "ft"
    base.mkdir()
    ft.mkdir()
    save_file({"a.weight": torch.ones(2, 3), "b.weight": torch.full((4,), 2.0)}, str(base / "ema.safetensors"))
    assert find_weights_file(str(base)).endswith("ema.safetensors")
    save_file({"a.weight": torch.ones(2, 3, dtype=torch.bfloat16) * 3, "b.weight": torch.full((4,), 2.0, dtype=torch.bfloat16)},
              str(base / "ema_bf16.safetensors"))
    assert find_weights_file(str(base)).endswith("ema_bf16.safetensors")
    save_file({"a.weight": torch.full((2, 3), 7.0)}, str(ft / "ema.safetensors"))
    shapes = {"a.weight": (2, 3), "b.weight": (4,)}
    get = checkpoint_getter(str(base), shapes, checkpoint_weight_path=str(ft))
    assert float(get("a.weight")[0, 0]) == 7.0          # fine-tuned tensor wins
    assert float(get("b.weight")[0]) == 2.0             # missing there -> base
    with pytest.raises(KeyError):
        get("c.weight")
    with pytest.raises(ValueError, match="shape"):
        SafetensorsGetter(str(base / "ema.safetensors"), {"a.weight": (3, 2)})("a.weight")
    with pytest.raises(FileNotFoundError):
        find_weights_file(str(tmp_path / "nothing"))
    # the reference's conversion utility (interactive_vqa_inferencer.py:93-114), kept as a method of both entry-point classes
    from safetensors.torch import load_file
    from unimedvl_amd.interactive_image_generator import ImageGenerator
    from unimedvl_amd.interactive_vqa_inferencer import VQAInferencer
    src, dst = str(ft / "ema.safetensors"), str(ft / "ema_bf16.safetensors")
    assert VQAInferencer().convert_checkpoint_to_bf16(str(ft / "missing.safetensors"), dst) is False
    assert VQAInferencer().convert_checkpoint_to_bf16(src, dst) is True
    out = load_file(dst)
    assert out["a.weight"].dtype == torch.bfloat16 and float(out["a.weight"][1, 2]) == 7.0
    dst2 = str(ft / "copy_bf16.safetensors")
    assert ImageGenerator().convert_checkpoint_to_bf16(dst, dst2) is True          # already bf16: copied
    assert load_file(dst2)["a.weight"].dtype == torch.bfloat16


def test_kernel_policy_queries_need_no_gpu():
    """umv_gemm_tile_config / umv_attn_prefill_tq are host-only: the shape -> kernel policy the GPU branch tests pin
    (tests/test_kernel_branches_gpu.py) can be read on a machine without a GPU"""
    from unimedvl_amd import _lib
    lib = _lib.load()
    # the bench's prefill / ViT / flow shapes go to the hand-interleaved 256x256 tile
    for M, N, K in ((8208, 4608, 3584), (8208, 37888, 3584), (8208, 3584, 18944), (8192, 3456, 1152), (2064, 37888, 3584)):
        assert lib.umv_gemm_tile_config(M, N, K) == 266, (M, N, K)
    assert lib.umv_gemm_tile_config(8192, 1152, 4304) == 288        # ViT fc2 / out-proj: 4 x 288 columns x 64 row blocks = 256 tiles
    assert lib.umv_gemm_tile_config(2048, 4608, 3584) == 288        # guided flow pass q/k/v
    assert lib.umv_gemm_tile_config(32768, 1152, 4304) == 384       # 32 images: 768 tiles of 384 x 128 = 3 full rounds
    assert lib.umv_gemm_tile_config(1026, 4608, 3584) == 268        # one image span: 256(n) x 128(m)
    assert lib.umv_gemm_tile_config(1026, 3584, 3584) == 270        # 128 x 128, two workgroups per CU
    assert lib.umv_gemm_tile_config(300, 1152, 608) == 64           # short K
    assert lib.umv_gemm_tile_config(8, 37888, 3584) == 0            # M <= 64: weight-streaming kernels
    # few rows on a wide N (round 4): 256 x 128 tiles when they cost fewer rounds x area than 384 x 128 / 256 x 256
    assert lib.umv_gemm_tile_config(272, 37888, 3584) == 268        # 8 x 34-token question prefill: 444 tiles instead of 297 of 384 x 128
    assert lib.umv_gemm_tile_config(128, 37888, 3584) == 268        # 128-sample decode gate/up: 148 tiles instead of 99
    assert lib.umv_gemm_tile_config(130, 37888, 3584) == 384        # 2 row blocks: 198 tiles of 384 x 128 stay one round
    assert lib.umv_gemm_tile_config(512, 37888, 3584) == 384        # a single image's guided flow pass (unchanged)
    assert lib.umv_gemm_tile_config(1024, 37888, 3584) == 266 and lib.umv_gemm_tile_config(2048, 37888, 3584) == 266
    # prefill attention: two q-tiles per wave from 512 workgroups on
    assert lib.umv_attn_prefill_tq(8, 28, 4, 128, 1026) == 2
    assert lib.umv_attn_prefill_tq(1, 28, 4, 128, 1026) == 1
    assert lib.umv_attn_prefill_tq(8, 16, 16, 72, 1024) == 2
    assert lib.umv_attn_prefill_tq(8, 28, 4, 128, 1) == 0           # decode: the per-wave kernel
    assert lib.umv_attn_prefill_tq(8, 2, 1, 64, 1000) == 0          # head dims without an LDS-shared kernel


def test_w4_kernels_own_their_agprs(tmp_path):
    """gemm_w4.hip keeps its accumulators in LITERAL AGPRs the compiler never sees.  That is only sound while the compiler does not touch
    AGPRs on its own while they hold accumulators - and hipcc spills VGPRs INTO AGPRs when a kernel needs more than 256 of them (it happened
    in an experiment: lane constants of the epilogue hoisted out of a persistent tile loop; results were garbage and piece offsets came back
    corrupted).  Compile the file to ISA (no GPU needed) and check every kernel: exactly NACC zeroing writes (`v_accvgpr_write_b32 aN, 0`);
    no compiler spill (`v_accvgpr_write_b32 aN, vM`) before the last MFMA nor before the kernel's last own accumulator read (literal `a[N]`),
    and none at all in a kernel whose epilogue leaves the AGPRs in more than one chunk (TM / JC > 1, JC = 2 for TN > 8 else 4: a later
    chunk's accumulators are still in AGPRs while the first chunk's epilogue runs)."""
    import re
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "unimedvl_amd", "csrc", "gemm_w4.hip")
    out = str(tmp_path / "w4.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                        "-o", out, src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = open(out).read()
    names = re.findall(r"^(_Z14gemm_w4_kernel\w+):", s, re.M)
    assert len(names) >= 6
    for n in names:
        a = s.index(n + ":")
        b = s.index(".Lfunc_end", a)
        body = s[a:b]
        m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E", n)
        tn, tm = int(m.group(3)), int(m.group(4))
        nacc = tn * tm * 4
        scratch = int(re.compile(r"; ScratchSize: (\d+)").search(s, b).group(1))
        zero_writes = len(re.findall(r"v_accvgpr_write_b32 a\[(?:0x[0-9a-f]+|\d+)\], 0\s", body))
        spills = [mm.start() for mm in re.finditer(r"v_accvgpr_write_b32 a\d+, v\d+", body)]
        last_mfma = max(mm.start() for mm in re.finditer(r"v_mfma", body))
        assert zero_writes == nacc, f"{n}: {zero_writes} zeroing writes for {nacc} accumulators"
        assert all(p > last_mfma for p in spills), f"{n}: the compiler spills into AGPRs while the k loop runs"
        # the kernel's own reads are the literal a[N] form (inline asm); a compiler spill write in front of the LAST of them would land on
        # accumulators a later epilogue chunk still has to read out (chunks = TM / (TN > 8 ? 2 : 4): the 384 x 128 kernel has two as well)
        own_reads = [mm.start() for mm in re.finditer(r"v_accvgpr_read_b32 v\d+, a\[(?:0x[0-9a-f]+|\d+)\]", body)]
        assert len(own_reads) == nacc, f"{n}: {len(own_reads)} literal accumulator reads for {nacc} accumulators"
        assert all(p > own_reads[-1] for p in spills), f"{n}: {len(spills)} compiler spills into AGPRs, some before the last accumulator was read out"
        chunks = tm // (2 if tn > 8 else 4)
        assert chunks == 1 or not spills, f"{n}: {len(spills)} compiler spills into AGPRs in a kernel whose epilogue has {chunks} chunks"
        assert scratch == 0 and "v_accvgpr_mov" not in body, f"{n}: scratch {scratch}"


def test_attention_prefill_isa_has_no_scratch(tmp_path):
    """The prefill attention kernels wait for their LDS-DMA pieces with hand-counted `s_waitcnt vmcnt(N)`; scratch traffic counts in vmcnt
    too, so a build that spills registers inside the key loop gives WRONG results (round 5: the lazy-softmax hd 72 kernel held to 128 VGPRs
    in a two-tiles-per-call form spilled 7 registers and failed on the GPU).  Compile the file to ISA (no GPU needed) and check every
    kernel: no scratch, no compiler use of AGPRs as spill space, and the occupancy the launch policy counts on (hd 72 / TQ = 2: 4 waves
    per SIMD, hd 128 / TQ = 2: 3)."""
    import re
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "unimedvl_amd", "csrc", "attention_prefill.hip")
    out = str(tmp_path / "ap.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form",
                        "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = open(out).read()
    names = re.findall(r"^(_Z19attn_prefill_kernelILi(\d+)ELi(\d+)ELi(\d)ELb(\d)ELb(\d)EEv\w+):", s, re.M)
    # hd 128 / 72 x TQ 1 / 2 x exact / lazy; the four counting (stats) instantiations of the lazy kernels; the two paged hd-128 lazy kernels
    assert len(names) == 14, names
    for n, hd, tq, lazy, stats, paged in names:
        b = s.index(".Lfunc_end", s.index(n + ":"))
        scratch = int(re.compile(r"; ScratchSize: (\d+)").search(s, b).group(1))
        occ = int(re.compile(r"; Occupancy: (\d+)").search(s, b).group(1))
        agprs = int(re.compile(r"; NumAgprs: (\d+)").search(s, b).group(1))
        assert scratch == 0 and agprs == 0, f"{n}: scratch {scratch}, AGPRs {agprs}"
        if tq == "2" and stats == "0":
            assert occ >= (4 if hd == "72" else 3), f"{n}: occupancy {occ}"
        if lazy != "0":
            # the lazy softmax is compiler-visible throughout: no inline asm next to its v_permlane*_swap steps (round 6)
            body = s[s.index(n + ":"):b]
            assert not re.search(r";;#ASMSTART\s+v_", body), f"{n}: inline-asm VALU instruction in a lazy kernel"
