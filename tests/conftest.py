import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
