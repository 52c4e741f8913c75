import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 1234          # must match tools/make_golden.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a machine without a GPU.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session", autouse=True)
def _threads():
    torch.set_num_threads(min(8, os.cpu_count() or 1))


def gen_state_dict(template_sd, g7):
    """GenNet weights of the G7 fixtures: synthetic (seeded), prior restricted to the K=128 codebooks, object codebook = the
    reference's object-type features of 128 seed clouds as stored in the fixture (tools/make_golden.py:gen_state_dict)."""
    from dvqvae_amd import synth
    sd = synth.synthetic_state_dict(template_sd, SEED)
    sd["GatedPixelCNN.output_conv.2.bias"][128:] = -1e4
    sd[synth.OBJECT_CODEBOOK] = torch.from_numpy(g7["E6_f16"].astype(np.float32))
    return sd


def dvqvae_state_dict(template_sd, g8):
    """DVQVAE weights of the G8 fixture: synthetic (seeded) with the seven codebooks stored in the fixture."""
    from dvqvae_amd import synth
    sd = synth.synthetic_state_dict(template_sd, SEED + 8)
    for k in range(7):
        sd[f"vqvae{k}.vector_quantization.embedding.weight"] = torch.from_numpy(g8[f"E{k}_f16"].astype(np.float32))
    return sd
