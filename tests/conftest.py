import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# GPU tests of experimental/ (kernels that were measured and NOT adopted; nothing in unimedvl_amd/ imports that package).  They cost the
# round-end GPU suite ~40 % of its time for code that is not the product: opt-in with UMV_TEST_EXPERIMENTAL=1 (+ python -m experimental.build).
EXPERIMENTAL_MODULES = {"test_attn_decode_fused_gpu", "test_attn_prefill32_gpu", "test_decode_engine_gpu", "test_gemm_decode_gpu"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: exercises experimental/ (opt-in: UMV_TEST_EXPERIMENTAL=1)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("UMV_TEST_EXPERIMENTAL", "0") not in ("0", ""):
        return
    skip = pytest.mark.skip(reason="experimental/ kernels are opt-in: UMV_TEST_EXPERIMENTAL=1 (and python -m experimental.build)")
    for item in items:
        if item.module.__name__.split(".")[-1] in EXPERIMENTAL_MODULES:
            item.add_marker(pytest.mark.experimental)
            item.add_marker(skip)


def load_golden(name):
    """npz -> dict of torch tensors; keys ending __bf16 are uint16 bit patterns."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if k.endswith("__bf16"):
            out[k[:-6]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        elif a.dtype.kind in "US":
            out[k] = str(a)
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def tiny_weights():
    from oracle.weights import TINY, make_weights, digest
    sd, vae_sd = make_weights(TINY)
    return dict(TINY), sd, vae_sd, digest(sd) + ":" + digest(vae_sd)


NEW_TOKEN_IDS = dict(bos_token_id=300, eos_token_id=301, start_of_image=302, end_of_image=303)
